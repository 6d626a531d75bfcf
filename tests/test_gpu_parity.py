"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the
oracle on the same seeded inputs and against the committed golden vectors.  Bit-exact
(integer work)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest

import __graft_entry__ as ge
import pssbam_testlib as tl

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
MANIFEST = json.loads((GOLD / "manifest.json").read_text())


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_pkg()
    assert p.LIB_HIP.exists(), "libpssbam_hip.so missing: the HIP path must be built, there is no fallback"
    return p


def _engine_tables(pkg, contigs, refs, raw, pss=None, kmer=None, rg=None, kernel=0, chunks=1):
    eng = pkg.Engine(pss=pss, kmer=kmer, read_group=rg, kernel=kernel)
    try:
        eng.set_genome_arrays(tl.loaded_contigs(contigs))
        eng.set_references([n for n, _ in refs])
        offs = pkg.index_records(raw)
        n = offs.size - 1
        cuts = [n * i // chunks for i in range(chunks + 1)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                o = offs[a:b + 1].astype(np.int64)
                eng.submit(raw[o[0]:o[-1]], (o - o[0]).astype(np.uint32))
        return eng.finish()
    finally:
        eng.close()


def _pss_dict(o: tl.PssOpts):
    return dict(region_len=o.region_len, min_read_len=o.min_read_len, max_read_len=o.max_read_len, min_mq=o.min_mq,
                up_ctx=o.up_ctx, down_ctx=o.down_ctx, merged_only=o.merged_only)


def _fk_dict(o: tl.FkOpts):
    return dict(klen=o.klen, min_mq=o.min_mq, min_read_len=o.min_read_len, max_read_len=o.max_read_len,
                merged_only=o.merged_only)


def _check_pss(got, want_fwd, want_rev, st):
    assert np.array_equal(got.fwd, want_fwd), "forward table differs"
    assert np.array_equal(got.rev, want_rev), "reverse table differs"
    assert got.stats["parse_skip"] == st[tl.ST_PARSE_SKIP]
    assert got.stats["no_contig"] == st[tl.ST_NO_CONTIG]
    assert got.stats["pss_ok"] == st[tl.ST_OK]
    assert got.stats["pss_filtered"] == st[tl.ST_FILTERED]


@pytest.mark.parametrize("seed", range(int(os.environ.get("PSSBAM_FUZZ_SEEDS", "8"))))   # more seeds for a soak run
def test_fuzz_both_kernels_match_oracle(pkg, oracle, tmp_path, seed):
    """random records x random options, text for the oracle / binary for the engine, the
    two encodings written independently from the same record list"""
    contigs, refs, recs = tl.fuzz_dataset(300 + seed, 2500)
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    raw = tl.raw_records(refs, recs)
    g = oracle.load_genome(fa)
    rng = np.random.default_rng(seed)
    try:
        for trial in range(4):
            po = tl.PssOpts() if trial == 0 else tl.random_pss_opts(rng)
            ko = tl.FkOpts(klen=4) if trial == 0 else tl.random_fk_opts(rng)
            wf, wr, st = oracle.pss(g, sam, po)
            w5, w3, stk = oracle.fragkon(g, sam, ko)
            kernels = [pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED]   # the tiled kernel takes a large -r in passes of 32 rows
            for kern in kernels:
                got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), kmer=_fk_dict(ko), kernel=kern,
                                     chunks=1 + trial)
                _check_pss(got, wf, wr, st)
                assert np.array_equal(got.k5, w5.astype(np.uint64)), f"5' k-mers differ ({ko})"
                assert np.array_equal(got.k3, w3.astype(np.uint64)), f"3' k-mers differ ({ko})"
                assert got.stats["kmer_ok"] == stk[tl.ST_OK]
                assert got.stats["kmer_filtered"] == stk[tl.ST_FILTERED]
                assert got.stats["kmer_fail"] == stk[tl.ST_KMER_FAIL]
                assert got.stats["records"] == len(recs)
    finally:
        oracle.free_genome(g)


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "pss-bam"],
                         ids=lambda c: c["prefix"])
def test_golden_pss_from_bam(pkg, case):
    """the committed BAM fixture through the engine == the reference's committed output"""
    ds = MANIFEST["datasets"][case["dataset"]]
    o = tl.PssOpts(**case["opts"])
    refs, raw = tl.read_bam(GOLD / ds["bam"])
    fa_txt = (GOLD / ds["fasta"]).read_text()
    contigs = [(blk.split("\n", 1)[0].split()[0], "".join(blk.split("\n")[1:])) for blk in fa_txt.split(">")[1:]]
    wf, wr = tl.parse_counts_text((GOLD / case["counts"]).read_text())
    for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
        got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(o), rg=o.read_group, kernel=kern)
        assert np.array_equal(got.fwd, wf) and np.array_equal(got.rev, wr)


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "fragkon"],
                         ids=lambda c: c["stdout"])
def test_golden_fragkon_from_bam(pkg, case):
    ds = MANIFEST["datasets"][case["dataset"]]
    o = tl.FkOpts(**case["opts"])
    refs, raw = tl.read_bam(GOLD / ds["bam"])
    fa_txt = (GOLD / ds["fasta"]).read_text()
    contigs = [(blk.split("\n", 1)[0].split()[0], "".join(blk.split("\n")[1:])) for blk in fa_txt.split(">")[1:]]
    w5, w3 = tl.parse_fragkon_text((GOLD / case["stdout"]).read_text())
    for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
        got = _engine_tables(pkg, contigs, refs, raw, kmer=_fk_dict(o), kernel=kern)
        assert np.array_equal(np.minimum(got.k5, 0xFFFFFFFF).astype(np.uint32), w5)
        assert np.array_equal(np.minimum(got.k3, 0xFFFFFFFF).astype(np.uint32), w3)


def test_read_group_filter(pkg, oracle, tmp_path):
    contigs, refs, recs = tl.fuzz_dataset(901, 2000, with_rg=True)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    raw = tl.raw_records(refs, recs)
    g = oracle.load_genome(fa)
    try:
        for rg in ("grpA", "grpB", "grp", "nope"):
            keep = [r for r in recs if ("RG", "Z", rg) in r.tags]
            sam = tmp_path / f"{rg}.sam"
            tl.write_sam(sam, refs, keep)
            po = tl.PssOpts(region_len=12)
            wf, wr, st = oracle.pss(g, sam, po)
            for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
                got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), rg=rg, kernel=kern)
                _check_pss(got, wf, wr, st)
                assert got.stats["rg_dropped"] == len(recs) - len(keep)
    finally:
        oracle.free_genome(g)


def test_edges_empty_ragged_accumulate_reset(pkg, oracle, tmp_path):
    contigs, refs, recs = tl.fuzz_dataset(55, 700)
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    g = oracle.load_genome(fa)
    wf, wr, _ = oracle.pss(g, sam, tl.PssOpts(region_len=10))
    oracle.free_genome(g)
    raw = tl.raw_records(refs, recs)
    offs = pkg.index_records(raw)
    eng = pkg.Engine(pss=dict(region_len=10))
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references([n for n, _ in refs])
    eng.submit(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint32))      # empty block
    z = eng.finish()
    assert z.fwd.sum() == 0 and z.stats["records"] == 0
    # ragged: blocks of 1, 2, 3, ... records
    a, step = 0, 1
    n = offs.size - 1
    while a < n:
        b = min(n, a + step)
        o = offs[a:b + 1].astype(np.int64)
        eng.submit(raw[o[0]:o[-1]], (o - o[0]).astype(np.uint32))
        a, step = b, step + 1
    once = eng.finish()
    assert np.array_equal(once.fwd, wf) and np.array_equal(once.rev, wr)
    eng.submit(raw, offs)                                                        # accumulates
    twice = eng.finish()
    assert np.array_equal(twice.fwd, 2 * wf) and np.array_equal(twice.rev, 2 * wr)
    eng.reset()
    assert eng.finish().fwd.sum() == 0
    eng.close()


def test_tile_overflow_path(pkg, oracle, tmp_path, monkeypatch):
    """records whose needed prefix exceeds what the tiled kernel stages in LDS take its
    out-of-line global-memory path; force that with a tiny prefix"""
    contigs, refs, recs = tl.fuzz_dataset(77, 1500)
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    g = oracle.load_genome(fa)
    po, ko = tl.PssOpts(region_len=7), tl.FkOpts(klen=3)
    wf, wr, st = oracle.pss(g, sam, po)
    w5, w3, _ = oracle.fragkon(g, sam, ko)
    oracle.free_genome(g)
    raw = tl.raw_records(refs, recs)
    monkeypatch.setenv("PSSBAM_TILE_READS", "64")
    monkeypatch.setenv("PSSBAM_PIECES", "5")     # 80 bytes per record in LDS: most records spill
    got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), kmer=_fk_dict(ko), kernel=pkg.KERNEL_TILED)
    _check_pss(got, wf, wr, st)
    assert np.array_equal(got.k5, w5.astype(np.uint64)) and np.array_equal(got.k3, w3.astype(np.uint64))


def test_large_region_len(pkg, oracle, tmp_path):
    """-r beyond one 32-row pass: the tiled kernel (AUTO's choice for every N) walks the block once
    per 32 table rows"""
    contigs, refs, recs = tl.fuzz_dataset(88, 1500)
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    g = oracle.load_genome(fa)
    raw = tl.raw_records(refs, recs)
    try:
        for n in (31, 62, 63, 64, 100, 200, 254, 255, 1100):
            po = tl.PssOpts(region_len=n)
            wf, wr, st = oracle.pss(g, sam, po)
            for kern in (pkg.KERNEL_AUTO, pkg.KERNEL_SIMPLE):
                got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), kmer=dict(klen=5), kernel=kern, chunks=3)
                _check_pss(got, wf, wr, st)
                assert got.stats["records"] == len(recs)
    finally:
        oracle.free_genome(g)


def test_c_abi_error_conventions(pkg, tmp_path):
    """0 = ok, negative PSSBAM_E* + a message from pssbam_last_error(); the library never exits
    (INTEGRATION.md): wrong call order, bad options, malformed blocks"""
    contigs, refs, recs = tl.fuzz_dataset(5, 50)
    raw = np.frombuffer(tl.raw_records(refs, recs), dtype=np.uint8)
    with pytest.raises(pkg.PssbamError, match="tally_mask"):
        pkg.Engine()                                              # neither tally requested
    with pytest.raises(pkg.PssbamError, match="klen"):
        pkg.Engine(kmer=dict(klen=16))
    with pytest.raises(pkg.PssbamError, match="region_len"):
        pkg.Engine(pss=dict(region_len=-1))
    with pytest.raises(pkg.PssbamError, match="does not exist"):
        pkg.Engine(pss=dict(region_len=5), device=99)
    eng = pkg.Engine(pss=dict(region_len=5))
    try:
        with pytest.raises(pkg.PssbamError, match="set_genome"):
            eng.submit(raw)                                       # no genome yet
        with pytest.raises(pkg.PssbamError, match="set_genome must precede"):
            eng.set_references([r[0] for r in refs])
        eng.set_genome_arrays(tl.loaded_contigs(contigs))
        with pytest.raises(pkg.PssbamError, match="set_references"):
            eng.submit(raw)                                       # no reference table yet
        eng.set_references([r[0] for r in refs])
        offs = pkg.index_records(raw)
        bad = offs.copy()
        bad[-1] -= 1
        with pytest.raises(pkg.PssbamError, match="offsets"):
            eng.submit(raw, bad)                                  # index does not cover the block
        with pytest.raises(pkg.PssbamError, match="partial record"):
            eng.submit(raw[:-3])                                  # the Python wrapper's own framing check
        torn = raw.copy()
        torn[0:4] = np.frombuffer(np.uint32(7).tobytes(), dtype=np.uint8)   # block_size < 32
        with pytest.raises(pkg.PssbamError, match="block_size"):
            pkg.index_records(torn)
        eng.submit(raw, offs)                                     # and the engine is still usable
        assert eng.finish().stats["records"] == len(recs)
    finally:
        eng.close()


def test_kmer_lengths_beyond_twelve(pkg, oracle, tmp_path):
    """fragkon / genome-kmer-count accept any -k in the reference (fragkon.c:260, tree beyond 8 bases
    kmer.c:67-98); the device keeps 4^k 64-bit bins per table in HBM up to k = 15 (2 x 8.6 GB).
    k = 13 through both kernels and the genome census, k = 15 through the production kernel."""
    contigs, refs, recs = tl.fuzz_dataset(1313, 3000, contig_lens=(60000, 9000, 1500))
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    raw = tl.raw_records(refs, recs)
    g = oracle.load_genome(fa)
    try:
        for k, kernels in ((13, (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED)), (15, (pkg.KERNEL_TILED,))):
            ko = tl.FkOpts(klen=k)
            w5, w3, stk = oracle.fragkon(g, sam, ko)
            assert int(w5.sum()) > 500
            for kern in kernels:
                got = _engine_tables(pkg, contigs, refs, raw, pss=dict(region_len=15), kmer=_fk_dict(ko), kernel=kern, chunks=2)
                assert np.array_equal(got.k5, w5), f"5' k-mers differ at k={k}"
                assert np.array_equal(got.k3, w3), f"3' k-mers differ at k={k}"
                assert got.stats["kmer_ok"] == stk[tl.ST_OK] and got.stats["kmer_fail"] == stk[tl.ST_KMER_FAIL]
                del got
            del w5, w3
        want = oracle.genome_kmer_count(g, 13)
        eng = pkg.Engine(kmer=dict(klen=13))
        eng.set_genome_arrays(tl.loaded_contigs(contigs))
        assert np.array_equal(eng.genome_kmer_count(13), want)
        eng.close()
    finally:
        oracle.free_genome(g)


def test_prefix_sample_covers_late_long_records(pkg, oracle, tmp_path):
    """the tiled kernels stage a record prefix whose size is sampled per block; a block whose LATER
    records are longer than its first ones must still be sampled right (start, middle and end are
    looked at), so nothing lands on the one-lane slow path -- host blocks and device-resident blocks"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")          # device buffers for the submit_device leg, straight from the HIP runtime
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    contigs, refs, recs = tl.fuzz_dataset(2024, 12000, contig_lens=(30000, 8000))
    recs = sorted(recs, key=lambda r: len(r.seq))            # short reads first, 250-bp reads last
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    g = oracle.load_genome(fa)
    po = tl.PssOpts(region_len=20)
    wf, wr, st = oracle.pss(g, sam, po)
    oracle.free_genome(g)
    raw = tl.raw_records(refs, recs)
    offs = pkg.index_records(raw)
    for mode in ("host", "device"):
        eng = pkg.Engine(pss=_pss_dict(po), kernel=pkg.KERNEL_TILED)
        eng.set_genome_arrays(tl.loaded_contigs(contigs))
        eng.set_references([n for n, _ in refs])
        if mode == "host":
            eng.submit(raw, offs)
        else:
            padded = np.concatenate([raw, np.zeros(64, dtype=np.uint8)])
            d_r, d_o = C.c_void_p(), C.c_void_p()
            assert hip.hipMalloc(C.byref(d_r), padded.size) == 0 and hip.hipMalloc(C.byref(d_o), offs.size * 4) == 0
            assert hip.hipMemcpy(d_r, padded.ctypes.data, padded.size, 1) == 0 and hip.hipMemcpy(d_o, offs.ctypes.data, offs.size * 4, 1) == 0
            eng.submit_device(d_r.value, raw.size, d_o.value, offs.size - 1)
        got = eng.finish()
        eng.close()
        if mode == "device":
            hip.hipFree(d_r), hip.hipFree(d_o)
        _check_pss(got, wf, wr, st)
        assert got.stats["slow_path"] == 0, (mode, got.stats)


@pytest.mark.parametrize("region_len", [16, 17])
def test_compact_tiled_switch_at_sixteen(pkg, oracle, tmp_path, monkeypatch, region_len):
    """-r 16 is tally_compact's last position column (slot 15 of the half tables, second v_perm group), -r 17 the
    first tally_tiled case: both pinned against the oracle, with and without -R, and PSSBAM_COMPACT=0 as the
    cross-check of the switch itself (csrc/engine.hip: rows <= COMPACT_MAX_ROWS)"""
    contigs, refs, recs = tl.fuzz_dataset(1600 + region_len, 3000, with_rg=True)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    raw = tl.raw_records(refs, recs)
    g = oracle.load_genome(fa)
    try:
        for rg in (None, "grpA"):
            keep = recs if rg is None else [r for r in recs if ("RG", "Z", rg) in r.tags]
            sam = tmp_path / f"{rg}.sam"
            tl.write_sam(sam, refs, keep)
            for po in (tl.PssOpts(region_len=region_len), tl.PssOpts(region_len=region_len, min_mq=20, up_ctx="CT", down_ctx="ACGTN")):
                wf, wr, st = oracle.pss(g, sam, po)
                for compact in ("1", "0"):
                    monkeypatch.setenv("PSSBAM_COMPACT", compact)
                    for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
                        got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), rg=rg, kernel=kern, chunks=2)
                        _check_pss(got, wf, wr, st)
    finally:
        oracle.free_genome(g)


@pytest.mark.parametrize("seed", [41, 42])
def test_compact_plan_once_form(pkg, oracle, tmp_path, monkeypatch, seed):
    """tally_compact's switchable form that decodes + filters in ONE lane per read and hands the plan to the pair through
    LDS (PSSBAM_COMPACT_PLAN_ONCE=1; measured slower, off by default -- profiles/r03_ab_plan_once.txt): same tables and
    status tallies as the oracle, fragkon's k-mer tables included, tiles that are not full included (chunks)"""
    contigs, refs, recs = tl.fuzz_dataset(seed, 4000)
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    tl.write_fasta(fa, contigs)
    tl.write_sam(sam, refs, recs)
    raw = tl.raw_records(refs, recs)
    g = oracle.load_genome(fa)
    monkeypatch.setenv("PSSBAM_COMPACT_PLAN_ONCE", "1")
    try:
        for po in (tl.PssOpts(region_len=15), tl.PssOpts(region_len=9, min_mq=10, up_ctx="CT", down_ctx="AG")):
            wf, wr, st = oracle.pss(g, sam, po)
            for chunks in (1, 3):
                got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), kernel=pkg.KERNEL_TILED, chunks=chunks)
                _check_pss(got, wf, wr, st)
        ko = tl.FkOpts(klen=4)
        w5, w3, _ = oracle.fragkon(g, sam, ko)
        got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(tl.PssOpts(region_len=15)), kmer=_fk_dict(ko), kernel=pkg.KERNEL_TILED, chunks=2)
        assert np.array_equal(got.k5, w5) and np.array_equal(got.k3, w3)
    finally:
        oracle.free_genome(g)


def test_read_group_behind_large_aux_arrays(pkg, oracle, tmp_path):
    """B arrays of thousands of elements (every sub-type), H strings and floats in front of and behind RG:Z: the -R
    walk of the device decoders must step over all of them by type (reference: `samtools view -r`,
    pss-bam.c:150-155).  BAM only -- text of this size would overflow the reference's 2047-byte tag buffer
    (precondition P2), and the oracle never looks at tags: it gets the kept records without them"""
    contigs, refs, recs = tl.fuzz_dataset(2718, 1500, with_rg=True)
    rng = np.random.default_rng(5)
    subs = "cCsSiIf"
    for i, r in enumerate(recs):
        sub = subs[i % 7]
        n = [0, 1, 700, 5000][(i // 7) % 4]
        lo, hi = tl._B_RANGE.get(sub, (0, 1))
        vals = [float(np.float32(x)) for x in rng.normal(0, 9, n)] if sub == "f" else [int(x) for x in rng.integers(lo, hi, n)]
        big = ("ZB", "B", (sub, vals))
        r.tags.insert(0 if i % 2 else len(r.tags), big)
        if i % 5 == 0:
            r.tags.insert(0, ("XH", "H", "AB" * int(rng.integers(0, 400))))
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    raw = tl.raw_records(refs, recs)
    g = oracle.load_genome(fa)
    try:
        for rg in ("grpA", "grpB"):
            keep = [tl.Rec(**{**r.__dict__, "tags": []}) for r in recs if ("RG", "Z", rg) in r.tags]
            sam = tmp_path / f"{rg}.sam"
            tl.write_sam(sam, refs, keep)
            po = tl.PssOpts(region_len=20)
            wf, wr, st = oracle.pss(g, sam, po)
            for kern in (pkg.KERNEL_SIMPLE, pkg.KERNEL_TILED):
                got = _engine_tables(pkg, contigs, refs, raw, pss=_pss_dict(po), rg=rg, kernel=kern)
                _check_pss(got, wf, wr, st)
                assert got.stats["rg_dropped"] == len(recs) - len(keep)
    finally:
        oracle.free_genome(g)
