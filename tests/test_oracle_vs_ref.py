"""Pins the CPU restatement (oracle/liboracle.so) to the UNMODIFIED reference build
(oracle/_ref/*, compiled from /root/reference by oracle/Makefile) on randomised input.

Skipped when oracle/_ref is absent (a checkout that never saw /root/reference); the
golden-vector test (test_golden.py) pins the oracle in that case.
"""
import os

import numpy as np
import pytest

from pssbam_testlib import (FkOpts, PssOpts, fuzz_dataset, have_ref, random_fk_opts, random_pss_opts, ref_safe,
                            run_ref_fragkon, run_ref_gkc, run_ref_pss, write_fasta, write_sam)

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")


def _dataset(tmp_path, seed, n=1500, **kw):
    contigs, refs, recs = fuzz_dataset(seed, n, **kw)
    recs = ref_safe(recs)
    fa = tmp_path / f"g{seed}.fa"
    sam = tmp_path / f"a{seed}.sam"
    write_fasta(fa, contigs, width=int(np.random.default_rng(seed).choice([50, 60, 61, 80])))
    write_sam(sam, refs, recs)
    return fa, sam, recs, refs


@pytest.mark.parametrize("seed", range(int(os.environ.get("PSSBAM_FUZZ_SEEDS", "12"))))
def test_pss_restatement_matches_reference(tmp_path, oracle, seed):
    fa, sam, _, _ = _dataset(tmp_path, seed)
    rng = np.random.default_rng(1000 + seed)
    g = oracle.load_genome(fa)
    try:
        for trial in range(3):
            o = PssOpts() if trial == 0 else random_pss_opts(rng)
            fwd, rev, _ = oracle.pss(g, sam, o)
            rf, rr, ctext, rtext, _ = run_ref_pss(fa, sam, tmp_path / f"ref{trial}", o)
            assert np.array_equal(fwd, rf), f"fwd differs seed={seed} opts={o}"
            assert np.array_equal(rev, rr), f"rev differs seed={seed} opts={o}"
            # report writers: byte-exact files (same -F/-B/-o strings)
            oracle.write_reports(str(fa), str(sam), str(tmp_path / f"ref{trial}"), fwd, rev)
            assert (tmp_path / f"ref{trial}.pss.counts.txt").read_text() == ctext
            assert (tmp_path / f"ref{trial}.pss.rates.txt").read_text() == rtext
    finally:
        oracle.free_genome(g)


def test_pss_counts_are_nontrivial(tmp_path, oracle):
    """guards against a fuzz set that everything filters out of"""
    fa, sam, _, _ = _dataset(tmp_path, 77)
    g = oracle.load_genome(fa)
    fwd, rev, st = oracle.pss(g, sam, PssOpts(region_len=5))
    oracle.free_genome(g)
    assert st[0] > 100 and fwd.sum() > 500 and rev.sum() > 500
    assert st[1] > 0 and st[2] > 0 and st[3] > 0        # parse-skip, no-contig, filtered all occur


@pytest.mark.parametrize("seed", range(int(os.environ.get("PSSBAM_FUZZ_SEEDS", "12"))))
def test_fragkon_restatement_matches_reference(tmp_path, oracle, seed):
    contigs, refs, recs = fuzz_dataset(100 + seed, 1500)
    rng = np.random.default_rng(2000 + seed)
    fa = tmp_path / "g.fa"
    write_fasta(fa, contigs)
    g = oracle.load_genome(fa)
    try:
        for trial in range(3):
            o = FkOpts(klen=4) if trial == 0 else random_fk_opts(rng)
            # precondition P4 (oracle/pss_oracle.c): starts < k/2 are undefined in the reference
            keep = ref_safe(recs, o.klen)
            sam = tmp_path / f"a{trial}.sam"
            write_sam(sam, refs, keep)
            k5, k3, _ = oracle.fragkon(g, sam, o)
            r5, r3, _, _ = run_ref_fragkon(fa, sam, o)
            assert np.array_equal(k5, r5), f"5' differs seed={seed} opts={o}"
            assert np.array_equal(k3, r3), f"3' differs seed={seed} opts={o}"
    finally:
        oracle.free_genome(g)


def test_gz_fasta_and_O2_build_agree(tmp_path, oracle):
    contigs, refs, recs = fuzz_dataset(5, 800)
    recs = ref_safe(recs)
    fa, fagz, sam = tmp_path / "g.fa", tmp_path / "g.fa.gz", tmp_path / "a.sam"
    write_fasta(fa, contigs)
    write_fasta(fagz, contigs, gz=True)
    write_sam(sam, refs, recs)
    o = PssOpts(region_len=8)
    a = run_ref_pss(fa, sam, tmp_path / "p0", o)
    b = run_ref_pss(fagz, sam, tmp_path / "p1", o, variant="pss-bam.O2")
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    g = oracle.load_genome(fagz)
    fwd, rev, _ = oracle.pss(g, sam, o)
    oracle.free_genome(g)
    assert np.array_equal(fwd, a[0]) and np.array_equal(rev, a[1])


@pytest.mark.parametrize("k", [1, 3, 4, 7, 9])
def test_genome_kmer_count_restatement_matches_reference(tmp_path, oracle, k):
    contigs, _, _ = fuzz_dataset(40 + k, 10, contig_lens=(30000, 4000, 600, 50))
    fa = tmp_path / "g.fa"
    write_fasta(fa, contigs)
    g = oracle.load_genome(fa)
    got = oracle.genome_kmer_count(g, k)
    oracle.free_genome(g)
    want, _ = run_ref_gkc(fa, k)
    assert np.array_equal(got, want) and got.sum() > 20000
