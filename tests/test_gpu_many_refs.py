"""BAM headers with more references than the tiled kernel caches in LDS (REF_LDS_ENTRIES = 64,
csrc/record_decode.h RefsLdsCached::get): real human headers list 86-3366 names, and the
reference's find_seq accepts any count (fasta-genome-io.h:10, fasta-genome-io.c:202-213).
Reads land on refIDs 0, 63, 64, 65, 130, 199 (contigs present in the FASTA), on header names the
FASTA lacks, and on '*'; both kernels, with and without -R, through the C ABI and through the
command-line front end, against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
import pssbam_testlib as tl
from test_gpu_parity import _check_pss, _engine_tables, _fk_dict, _pss_dict

pytestmark = pytest.mark.gpu

SLOTS = [0, 63, 64, 65, 199, 130]   # header positions of the six real contigs
N_REF = 210


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_pkg()
    assert p.LIB_HIP.exists(), "libpssbam_hip.so missing: the HIP path must be built, there is no fallback"
    return p


def _dataset(seed, with_rg):
    contigs, _, recs = tl.fuzz_dataset(seed, 4000, contig_lens=(5000, 1200, 300, 900, 700, 2500), with_rg=with_rg)
    refs = [(f"unplaced_{i:03d}", 1000 + i) for i in range(N_REF)]
    for (nm, s), k in zip(contigs, SLOTS):
        refs[k] = (nm, len(s))
    refs[7] = ("chrMissing", 4000)
    rng = np.random.default_rng(seed)
    for r in recs:                    # some reads on header names the FASTA does not have, on either side of 64
        v = rng.random()
        if v < 0.02:
            r.rname = "unplaced_020"
        elif v < 0.04:
            r.rname = "unplaced_100"
        elif v < 0.05:
            r.rname = f"unplaced_{N_REF - 1:03d}"
    used = {r.rname for r in recs}
    assert {nm for nm, _ in contigs} <= used and "*" in used and "chrMissing" in used
    return contigs, refs, recs


@pytest.mark.parametrize("with_rg", [False, True], ids=["all_reads", "read_group"])
def test_header_with_210_references(pkg, oracle, tmp_path, with_rg):
    contigs, refs, recs = _dataset(4100 + int(with_rg), with_rg)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    raw = tl.raw_records(refs, recs)
    ids = np.frombuffer(raw, dtype=np.uint8)
    g = oracle.load_genome(fa)
    try:
        for rg in ([None] if not with_rg else ["grpA", "grpB"]):
            keep = recs if rg is None else [r for r in recs if ("RG", "Z", rg) in r.tags]
            sam = tmp_path / f"a_{rg}.sam"
            tl.write_sam(sam, refs, keep)
            for po, ko in ((tl.PssOpts(region_len=25), tl.FkOpts(klen=4)), (tl.PssOpts(region_len=40, min_mq=10), tl.FkOpts(klen=7))):
                wf, wr, st = oracle.pss(g, sam, po)
                w5, w3, stk = oracle.fragkon(g, sam, ko)
                assert wf.sum() > 1000
                for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
                    got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), kmer=_fk_dict(ko), rg=rg, kernel=kern,
                                         chunks=3)
                    _check_pss(got, wf, wr, st)
                    assert np.array_equal(got.k5, w5.astype(np.uint64)) and np.array_equal(got.k3, w3.astype(np.uint64))
                    assert got.stats["kmer_ok"] == stk[tl.ST_OK] and got.stats["kmer_fail"] == stk[tl.ST_KMER_FAIL]
                    assert got.stats["rg_dropped"] == len(recs) - len(keep)
                    assert got.stats["slow_path"] == 0 or kern == pkg.KERNEL_SIMPLE or rg is not None
    finally:
        oracle.free_genome(g)
    del ids


def test_cli_on_bam_with_210_references(pkg, oracle, tmp_path):
    """the same header through bin/pss-bam and bin/fragkon (BGZF BAM in, report files out)"""
    contigs, refs, recs = _dataset(4200, False)
    fa, sam, bam = tmp_path / "g.fa", tmp_path / "a.sam", tmp_path / "a.bam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    tl.write_bam(bam, refs, recs, level=1)
    g = oracle.load_genome(fa)
    po, ko = tl.PssOpts(region_len=18), tl.FkOpts(klen=5)
    wf, wr, st = oracle.pss(g, sam, po)
    w5, w3, _ = oracle.fragkon(g, sam, ko)
    oracle.free_genome(g)
    b = pkg.PKG_DIR / "bin"
    pr = subprocess.run([str(b / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / "out")] + po.argv(),
                        capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1"})
    assert pr.returncode == 0, pr.stderr
    gf, gr = tl.parse_counts_text((tmp_path / "out.pss.counts.txt").read_text())
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr)
    assert f"[pssbam] no_contig={st[tl.ST_NO_CONTIG]}" in pr.stderr
    pr = subprocess.run([str(b / "fragkon"), "-F", str(fa), "-B", str(bam)] + ko.argv(), capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    g5, g3 = tl.parse_fragkon_text(pr.stdout)
    assert np.array_equal(g5, w5) and np.array_equal(g3, w3)
