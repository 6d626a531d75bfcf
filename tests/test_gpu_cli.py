"""End-to-end on the GPU: the C front ends (bin/pss-bam, bin/fragkon) on BGZF BAM input must
write byte-for-byte what the reference wrote for the same records (golden vectors), and match
the oracle on a larger generated file."""
import json
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

import __graft_entry__ as ge
import pssbam_testlib as tl

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
MANIFEST = json.loads((GOLD / "manifest.json").read_text())


@pytest.fixture(scope="module")
def bins():
    pkg = ge.load_pkg()
    b = pkg.PKG_DIR / "bin"
    assert (b / "pss-bam").exists() and (b / "fragkon").exists(), "front ends not built (run __graft_entry__.build())"
    return b


def _stage(case, tmp_path):
    """the golden reports echo the -F/-B strings the reference was run with (set?.fa / set?.sam):
    give the BAM that very name so the files can be compared byte for byte"""
    ds = MANIFEST["datasets"][case["dataset"]]
    shutil.copy(GOLD / ds["fasta"], tmp_path / ds["fasta"])
    shutil.copy(GOLD / ds["bam"], tmp_path / ds["sam"])
    return ds


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "pss-bam"], ids=lambda c: c["prefix"])
def test_pss_cli_golden(bins, case, tmp_path):
    ds = _stage(case, tmp_path)
    o = tl.PssOpts(**case["opts"])
    pr = subprocess.run([str(bins / "pss-bam"), "-F", ds["fasta"], "-B", ds["sam"], "-o", case["prefix"]] + o.argv(),
                        cwd=tmp_path, capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1"})
    assert pr.returncode == 0, pr.stderr
    assert (tmp_path / case["counts"]).read_text() == (GOLD / case["counts"]).read_text()
    assert (tmp_path / case["rates"]).read_text() == (GOLD / case["rates"]).read_text()
    err = pr.stderr
    assert err.startswith("Full command: ") and "Reading genome sequence from:\n" in err
    assert "Finished loading genome.\nCounting matches/mismatches from:\n" in err and err.rstrip().endswith("Done.")
    assert f"[pssbam] records={ds['n_records']}" in err


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "fragkon"], ids=lambda c: c["stdout"])
def test_fragkon_cli_golden(bins, case, tmp_path):
    ds = _stage(case, tmp_path)
    o = tl.FkOpts(**case["opts"])
    pr = subprocess.run([str(bins / "fragkon"), "-F", ds["fasta"], "-B", ds["sam"]] + o.argv(), cwd=tmp_path,
                        capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    assert pr.stdout == (GOLD / case["stdout"]).read_text()


def test_cli_on_generated_bam_vs_oracle(bins, oracle, tmp_path):
    """120 k generated reads (C4-like: mixed lengths, clips, indels, damage) through BGZF + CLI,
    against the oracle on the model's independent SAM text; gz FASTA for the loader"""
    pkg = ge.load_pkg()
    from pss_bam_amd import synth
    d = synth.config("C4", n_reads=120_000, scale_genome=0.0005)
    region_len = d.pop("region_len")
    cfg = synth.make_cfg(**d)
    n = int(cfg.n_reads)
    recs, _ = synth.records_host(cfg, 0, n)
    names = [synth.contig_name(cfg, k) for k in range(int(cfg.n_contigs))]
    refs = [(names[k], int(cfg.contig_len[k])) for k in range(len(names))]
    import struct
    th = ("@HD\tVN:1.6\n" + "".join(f"@SQ\tSN:{a}\tLN:{b}\n" for a, b in refs)).encode()
    raw = b"BAM\1" + struct.pack("<i", len(th)) + th + struct.pack("<i", len(refs))
    for a, b in refs:
        raw += struct.pack("<i", len(a) + 1) + a.encode() + b"\0" + struct.pack("<i", b)
    raw += recs.tobytes()
    bam = tmp_path / "gen.bam"
    with open(bam, "wb") as fh:
        for i in range(0, len(raw), 0xFF00):
            fh.write(tl.bgzf_block(raw[i:i + 0xFF00], 1))
        fh.write(tl.BGZF_EOF)
    fa, sam = tmp_path / "gen.fa", tmp_path / "gen.sam"
    synth.fasta_host(cfg, fa)
    subprocess.run(["gzip", "-1", "-k", str(fa)], check=True)
    synth.sam_host(cfg, 0, n, sam)
    g = oracle.load_genome(fa)
    po, ko = tl.PssOpts(region_len=region_len, min_mq=20), tl.FkOpts(klen=6)
    wf, wr, st = oracle.pss(g, sam, po)
    w5, w3, _ = oracle.fragkon(g, sam, ko)
    oracle.free_genome(g)
    pr = subprocess.run([str(bins / "pss-bam"), "-F", str(fa) + ".gz", "-B", str(bam), "-o", str(tmp_path / "out")] + po.argv(),
                        capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    gf, gr = tl.parse_counts_text((tmp_path / "out.pss.counts.txt").read_text())
    assert np.array_equal(gf, wf) and np.array_equal(gr, wr) and st[tl.ST_OK] > 10000
    pr = subprocess.run([str(bins / "fragkon"), "-F", str(fa), "-B", str(bam)] + ko.argv(), capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    g5, g3 = tl.parse_fragkon_text(pr.stdout)
    assert np.array_equal(g5, w5) and np.array_equal(g3, w3)
    # the multi-engine path of the front ends (PSSBAM_NGPU: batches dealt round-robin to one engine
    # per GPU, counter blocks summed at the end), run here with three engines sharing the one GPU
    # and small batches so that every engine gets several; same tables, both ways of summing
    for reduce in ("", "host"):
        env = {**os.environ, "PSSBAM_NGPU": "3", "PSSBAM_OVERSUBSCRIBE": "1", "PSSBAM_BATCH_BYTES": str(1 << 20),
               "PSSBAM_REDUCE": reduce, "PSSBAM_STATS": "1", "PSSBAM_DEVICE_INFLATE": "0"}   # the HOST reader's asynchronous feed
        pr = subprocess.run([str(bins / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / "out3")] + po.argv(),
                            capture_output=True, text=True, env=env)
        assert pr.returncode == 0, pr.stderr
        assert "gpus=3" in pr.stderr
        gf, gr = tl.parse_counts_text((tmp_path / "out3.pss.counts.txt").read_text())
        assert np.array_equal(gf, wf) and np.array_equal(gr, wr)
        assert f"records={n}" in pr.stderr
        # the feed is asynchronous: runs of 2 batches per engine, 3 * 2 copies in flight, and the copy
        # windows of different engines overlap in time (they used to be strictly serial)
        import re
        m = re.search(r"feed: (\d+) batches in runs of (\d+) over 3 engines, up to (\d+) copies in flight; copy windows of "
                      r"different engines overlapped for ([\d.]+) s", pr.stderr)
        assert m, pr.stderr[-2000:]
        assert int(m.group(1)) >= 12 and int(m.group(2)) == 2 and int(m.group(3)) == 6
        wins = [(int(g), float(a), float(b)) for g, a, b in re.findall(r"copy window: engine (\d)  ([\d.]+) \.\. ([\d.]+) s", pr.stderr)]
        assert {g for g, _, _ in wins} == {0, 1, 2}
        assert [g for g, _, _ in wins[:6]] == [0, 0, 1, 1, 2, 2]          # contiguous runs, not round-robin
        assert any(a1 < b0 and a0 < b1 for (g0, a0, b0) in wins for (g1, a1, b1) in wins if g0 != g1), wins
        assert float(m.group(4)) >= 0      # (printed with millisecond resolution: 1 MiB copies overlap for microseconds)
    # and a run length of 1 / other geometries give the same tables
    for extra in ({"PSSBAM_RUN_BATCHES": "1"}, {"PSSBAM_RUN_BATCHES": "3", "PSSBAM_SLOTS": "4"}):
        env = {**os.environ, "PSSBAM_NGPU": "2", "PSSBAM_OVERSUBSCRIBE": "1", "PSSBAM_BATCH_BYTES": str(1 << 20),
               "PSSBAM_DEVICE_INFLATE": "0", **extra}
        pr = subprocess.run([str(bins / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / "out4")] + po.argv(),
                            capture_output=True, text=True, env=env)
        assert pr.returncode == 0, pr.stderr
        gf, gr = tl.parse_counts_text((tmp_path / "out4.pss.counts.txt").read_text())
        assert np.array_equal(gf, wf) and np.array_equal(gr, wr)


def test_reduce_counters_single_engine_is_identity(tmp_path):
    import ctypes as C
    pkg = ge.load_pkg()
    eng = pkg.Engine(pss=dict(region_len=5))
    arr = (C.c_void_p * 1)(eng._h)
    L = pkg.hip_lib()
    L.pssbam_reduce_counters.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
    assert L.pssbam_reduce_counters(arr, 1, 0) == 0
    assert L.pssbam_reduce_counters(arr, 1, 3) != 0      # bad root
    eng.close()


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] != "genome-kmer-count"],
                         ids=lambda c: c.get("prefix") or c["stdout"])
def test_cli_reads_sam_text_like_the_reference_pipeline(bins, case, tmp_path):
    """the very SAM files the reference was run on (plain, and gzipped) straight into the front
    ends: byte-identical reports, including every -R case"""
    ds = MANIFEST["datasets"][case["dataset"]]
    shutil.copy(GOLD / ds["fasta"], tmp_path / ds["fasta"])
    shutil.copy(GOLD / ds["sam"], tmp_path / ds["sam"])
    if case["tool"] == "pss-bam":
        o = tl.PssOpts(**case["opts"])
        pr = subprocess.run([str(bins / "pss-bam"), "-F", ds["fasta"], "-B", ds["sam"], "-o", case["prefix"]] + o.argv(),
                            cwd=tmp_path, capture_output=True, text=True)
        assert pr.returncode == 0, pr.stderr
        assert (tmp_path / case["counts"]).read_text() == (GOLD / case["counts"]).read_text()
        assert (tmp_path / case["rates"]).read_text() == (GOLD / case["rates"]).read_text()
    else:
        o = tl.FkOpts(**case["opts"])
        pr = subprocess.run([str(bins / "fragkon"), "-F", ds["fasta"], "-B", ds["sam"]] + o.argv(), cwd=tmp_path,
                            capture_output=True, text=True)
        assert pr.returncode == 0, pr.stderr
        assert pr.stdout == (GOLD / case["stdout"]).read_text()
        subprocess.run(["gzip", "-k", str(tmp_path / ds["sam"])], check=True)
        pr = subprocess.run([str(bins / "fragkon"), "-F", ds["fasta"], "-B", ds["sam"] + ".gz"] + o.argv(), cwd=tmp_path,
                            capture_output=True, text=True)
        assert pr.returncode == 0, pr.stderr
        assert pr.stdout.replace(ds["sam"] + ".gz", ds["sam"]) == (GOLD / case["stdout"]).read_text()


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "genome-kmer-count"], ids=lambda c: c["stdout"])
def test_genome_kmer_count_cli_golden(bins, case, tmp_path):
    ds = MANIFEST["datasets"][case["dataset"]]
    shutil.copy(GOLD / ds["fasta"], tmp_path / ds["fasta"])
    pr = subprocess.run([str(bins / "genome-kmer-count"), "-f", ds["fasta"], "-k", str(case["klen"])], cwd=tmp_path,
                        capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    assert pr.stdout == (GOLD / case["stdout"]).read_text()


def test_genome_kmer_count_engine_vs_oracle(oracle, tmp_path):
    pkg = ge.load_pkg()
    rng = np.random.default_rng(3)
    contigs = [(f"c{i}", tl.random_contig(rng, n)) for i, n in enumerate((250_000, 70_001, 513, 9, 3))]
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    g = oracle.load_genome(fa)
    eng = pkg.Engine(kmer=dict(klen=4))
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12):   # <= 8: packed genome + LDS histogram (two passes at 8); above: global bins
        want = oracle.genome_kmer_count(g, k)
        assert np.array_equal(eng.genome_kmer_count(k), want.astype(np.uint64)), k
    os.environ["PSSBAM_GKC_BYTES"] = "1"            # the byte-genome kernel stays as the cross-check
    try:
        for k in (4, 8):
            assert np.array_equal(eng.genome_kmer_count(k), oracle.genome_kmer_count(g, k).astype(np.uint64)), k
    finally:
        del os.environ["PSSBAM_GKC_BYTES"]
    eng.close()
    oracle.free_genome(g)
