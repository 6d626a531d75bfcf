"""Device-side BGZF inflate (csrc/inflate_kernels.h): stored, fixed-Huffman and
dynamic-Huffman deflate blocks inflated on the GPU must be byte-identical to zlib's output, the
per-block ISIZE / CRC-32 check must pass on good input and flag damaged input, and nothing may
fault on malformed streams.  Oracle: zlib (and, through the CRC in every BGZF trailer, the writer)."""
import struct
import zlib
from pathlib import Path

import numpy as np
import pytest

import __graft_entry__ as ge
import pssbam_testlib as tl

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(params=["in-place", "one-wait-per-step", "one-wait-whole-copies", "wave-per-block"], autouse=True)
def inflate_loop(request, monkeypatch):
    """every test of this module runs under every inflate path: the lane-per-block kernel's data loops (csrc/inflate_kernels.h:
    in place; one wait per step with copies moved in pieces of <= 64 bytes; the same with whole copies) and the
    wave-per-block pair of kernels (csrc/inflate_wave.h), which is the feed's default"""
    monkeypatch.setenv("PSSBAM_INFLATE_LOOP", "0" if request.param == "in-place" else "1")
    monkeypatch.setenv("PSSBAM_INFLATE_PIECES", "0" if request.param == "one-wait-whole-copies" else "1")
    monkeypatch.setenv("PSSBAM_INFLATE_WAVE", "1" if request.param == "wave-per-block" else "0")


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_pkg()
    assert p.LIB_HIP.exists(), "libpssbam_hip.so missing: the HIP path must be built, there is no fallback"
    return p


def _bgzf(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, split=None) -> bytes:
    """one BGZF block; split = (n, ...) emits several deflate blocks inside it (Z_FULL_FLUSH between)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    if split:
        parts, o = [], 0
        for n in split:
            parts.append(co.compress(data[o:o + n]) + co.flush(zlib.Z_FULL_FLUSH))
            o += n
        payload = b"".join(parts) + co.compress(data[o:]) + co.flush()
    else:
        payload = co.compress(data) + co.flush()
    bsize = len(payload) + 25
    assert bsize < 65536 + 26
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize) + payload
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def _payloads(rng):
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = [b"", b"x", b"ab" * 3, bytes(rng.integers(0, 256, 1000, dtype=np.uint8)),            # tiny / incompressible
           b"I" * 60000, b"abcdefg" * 9000, b"ab" * 30000, b"abc" * 21000, b"abcde" * 13000,     # distances 1..7
           acgt[rng.integers(0, 4, 65280)].tobytes(), bytes(rng.integers(0, 256, 65280, dtype=np.uint8)),
           bytes(rng.integers(0, 4, 65536, dtype=np.uint8)),                                     # a full 64 KiB block
           ("@read%07d\tchr1\t%d\n" * 1).encode() * 1]
    text = "".join(f"r{i:07d}\t{int(rng.integers(0, 1 << 20))}\tchr{int(rng.integers(1, 23))}\t{'I' * int(rng.integers(20, 150))}\n"
                   for i in range(700)).encode()
    out.append(text[:65000])
    skew = rng.choice(np.arange(256, dtype=np.uint8), size=64000, p=np.r_[[0.5], np.full(255, 0.5 / 255)])
    out.append(skew.tobytes())                                                                   # long and short codes
    return out


def test_stored_fixed_dynamic_blocks_match_zlib(pkg):
    rng = np.random.default_rng(11)
    blocks, want = [], []
    for data in _payloads(rng):
        for level, strategy, split in ((0, zlib.Z_DEFAULT_STRATEGY, None), (1, zlib.Z_DEFAULT_STRATEGY, None),
                                       (6, zlib.Z_DEFAULT_STRATEGY, None), (9, zlib.Z_DEFAULT_STRATEGY, None),
                                       (6, zlib.Z_FIXED, None), (6, zlib.Z_HUFFMAN_ONLY, None), (6, zlib.Z_RLE, None),
                                       (6, zlib.Z_DEFAULT_STRATEGY, (len(data) // 3, len(data) // 3)),
                                       (1, zlib.Z_FIXED, (len(data) // 2,))):
            if level == 0 and len(data) > 65000:
                continue     # a stored 64 KiB payload plus its framing exceeds the BGZF block size
            blocks.append(_bgzf(data, level, strategy, split))
            want.append(data)
    # every deflate block type is really present
    kinds = set()
    for b in blocks:
        if len(b) > 26:
            kinds.add((b[18] >> 1) & 3)
    assert kinds == {0, 1, 2}
    buf = np.frombuffer(b"".join(blocks) + tl.BGZF_EOF, dtype=np.uint8)
    res = pkg.bgzf_inflate(buf)
    assert res["bad_block"] is None, f"block {res['bad_block']} failed with status {res['bad_status']}"
    assert res["n_blocks"] == len(blocks) + 1
    assert res["data"].tobytes() == b"".join(want)


@pytest.mark.parametrize("name", ["setA.bam", "setB.bam"])
def test_golden_bams_inflate_like_zlib(pkg, name):
    raw = (GOLD / name).read_bytes()
    res = pkg.bgzf_inflate(np.frombuffer(raw, dtype=np.uint8))
    assert res["bad_block"] is None
    assert res["data"].tobytes() == tl.bgzf_inflate(raw)


def test_generated_bam_levels_and_layouts(pkg, tmp_path):
    """the synthetic BAM writer's files (htslib layout and blocks cut regardless of records), levels
    0 / 1 / 6: device output == zlib's, and the records index cleanly"""
    from pss_bam_amd import synth
    d = synth.config("C4", scale_genome=0.001, n_reads=150_000)
    d.pop("region_len")
    cfg = synth.make_cfg(**d)
    ref = None
    for level, ragged in ((1, False), (6, True), (0, False)):
        bam = tmp_path / f"g{level}{int(ragged)}.bam"
        synth.bam_file_host(cfg, 0, 150_000, bam, level=level, threads=4, ragged=ragged)
        raw = bam.read_bytes()
        res = pkg.bgzf_inflate(np.frombuffer(raw, dtype=np.uint8))
        assert res["bad_block"] is None, (level, ragged, res["bad_block"], res["bad_status"])
        got = res["data"].tobytes()
        assert got == tl.bgzf_inflate(raw)
        ref = ref or got
        assert got == ref            # same records whatever the level / layout


def test_damaged_blocks_are_flagged_not_followed(pkg):
    rng = np.random.default_rng(5)
    data = [("q%06d\t" % i).encode() * 40 + bytes(rng.integers(65, 70, 300, dtype=np.uint8)) for i in range(64)]
    blocks = [_bgzf(d, 6) for d in data]
    good = b"".join(blocks)
    for trial in range(40):
        k = int(rng.integers(0, len(blocks)))
        bad = bytearray(blocks[k])
        kind = trial % 4
        if kind == 0:     # payload bit flip
            o = int(rng.integers(18, len(bad) - 8))
            bad[o] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:   # CRC field
            bad[-8] ^= 0x40
        elif kind == 2:   # ISIZE too large (still <= 64 KiB)
            struct.pack_into("<I", bad, len(bad) - 4, min(65536, len(data[k]) + int(rng.integers(1, 200))))
        else:             # ISIZE too small
            struct.pack_into("<I", bad, len(bad) - 4, max(0, len(data[k]) - int(rng.integers(1, 200))))
        buf = b"".join(blocks[:k]) + bytes(bad) + b"".join(blocks[k + 1:])
        res = pkg.bgzf_inflate(np.frombuffer(buf, dtype=np.uint8))
        assert res["bad_block"] == k and res["bad_status"] != 0, (trial, kind, res["bad_block"], res["bad_status"])
    res = pkg.bgzf_inflate(np.frombuffer(good, dtype=np.uint8))
    assert res["bad_block"] is None and res["data"].tobytes() == b"".join(data)


def test_hostile_streams_never_fault(pkg):
    """random bytes dressed up as BGZF blocks: every block must come back flagged (or, by chance,
    valid), and the call must return"""
    rng = np.random.default_rng(99)
    blocks = []
    for i in range(256):
        n = int(rng.integers(1, 3000))
        payload = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        if i % 3 == 0:
            payload = bytes([0b101]) + payload      # BFINAL + dynamic header, garbage behind it
        isize = int(rng.integers(0, 65537))
        blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(payload) + 25) + payload
                      + struct.pack("<II", int(rng.integers(0, 1 << 32)), isize))
    res = pkg.bgzf_inflate(np.frombuffer(b"".join(blocks), dtype=np.uint8))
    assert res["n_blocks"] == 256 and res["bad_block"] is not None


def _bam_header_bytes(raw: bytes) -> int:
    """inflated bytes in front of the first alignment record"""
    data = tl.bgzf_inflate(raw)
    l_text = struct.unpack_from("<i", data, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", data, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, o)[0]
        o += 4 + l_name + 4
    return o


def _contigs_of(fa_path):
    fa_txt = Path(fa_path).read_text()
    return [(blk.split("\n", 1)[0].split()[0], "".join(blk.split("\n")[1:])) for blk in fa_txt.split(">")[1:]]


@pytest.mark.parametrize("seed,with_rg", [(41, False), (42, True)])
def test_feed_compressed_bam_to_tables(pkg, tmp_path, seed, with_rg):
    """submit_bgzf (copy compressed, inflate + CRC + record index + tally on the device) gives the same
    tables and status tallies as the host path fed the inflated records, for every kernel variant the
    device feed can use (N <= 16: tally_compact, N > 16: tally_tiled, k-mer tally fused, -R)"""
    contigs, refs, recs = tl.fuzz_dataset(seed, 4000, with_rg=with_rg)
    bam = tmp_path / "aligned.bam"
    hb = tl.write_bam_aligned(bam, refs, recs, level=1 + seed % 6, rng=np.random.default_rng(seed))
    raw = bam.read_bytes()
    assert hb == _bam_header_bytes(raw)
    rec_bytes = tl.raw_records(refs, recs)
    for pss, kmer, rg in ((dict(region_len=15), None, None), (dict(region_len=25, min_mq=10), dict(klen=4), None),
                          (dict(region_len=40), dict(klen=7), "grpA" if with_rg else None)):
        want = None
        for mode in ("host", "bgzf", "bgzf_small_batches"):
            eng = pkg.Engine(pss=pss, kmer=kmer, read_group=rg)
            eng.set_genome_arrays(tl.loaded_contigs(contigs))
            eng.set_references([n for n, _ in refs])
            if mode == "host":
                eng.submit(rec_bytes)
            else:
                eng.submit_bgzf(np.frombuffer(raw, dtype=np.uint8), header_bytes=hb,
                                max_batch_inflated=(1 << 30) if mode == "bgzf" else 70000)
                assert eng.feed_status()["flags"] == 0
            got = eng.finish()
            eng.close()
            if want is None:
                want = got
                assert got.stats["records"] == len(recs)
            else:
                assert np.array_equal(got.fwd, want.fwd) and np.array_equal(got.rev, want.rev), mode
                if kmer:
                    assert np.array_equal(got.k5, want.k5) and np.array_equal(got.k3, want.k3), mode
                st_g, st_w = dict(got.stats), dict(want.stats)
                st_g.pop("slow_path"), st_w.pop("slow_path")
                assert st_g == st_w, (mode, got.stats, want.stats)


def _feed_tables(pkg, contigs, refs, raw_bgzf, hb, pss, kmer=None, max_batch=1 << 30):
    eng = pkg.Engine(pss=pss, kmer=kmer)
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references([n for n, _ in refs])
    eng.submit_bgzf(np.frombuffer(raw_bgzf, dtype=np.uint8), header_bytes=hb, max_batch_inflated=max_batch)
    st = eng.feed_status()
    got = eng.finish()
    eng.close()
    return got, st


def test_submit_bgzf_rejects_a_wrong_block_table(pkg):
    """the kernels form addresses from the caller's table: one that lies is refused on the host"""
    import ctypes as C

    class Blk(C.Structure):
        _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint32), ("isize", C.c_uint32), ("out_off", C.c_uint64),
                    ("crc", C.c_uint32), ("status", C.c_uint32)]
    contigs = [("c1", "ACGT" * 500)]
    eng = pkg.Engine(pss=dict(region_len=5))
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references(["c1"])
    L = eng._L
    L.pssbam_engine_submit_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    comp = np.zeros(4096, dtype=np.uint8)

    def submit(rows):
        blocks = (Blk * len(rows))(*[Blk(*r, 0, 0) for r in rows])
        return L.pssbam_engine_submit_bgzf(eng._h, comp.ctypes.data, comp.size, blocks, len(rows), 0, None)
    good = [(18, 100, 500, 0), (150, 100, 500, 500)]
    for bad in ([(18, 100, 70000, 0)],                          # ISIZE above 64 KiB
                [(18, 70000, 500, 0)],                          # payload above 64 KiB
                [(4000, 200, 500, 0)],                          # payload runs past the chunk
                [(1 << 40, 10, 500, 0)],                        # payload nowhere near it
                [good[0], (150, 100, 500, 400)],                # out_off overlaps the previous block
                [good[0], (150, 100, 500, 600)],                # ... or leaves a hole
                [good[0], (100, 100, 500, 500)]):               # payloads out of file order
        assert submit(bad) == -1, bad          # PSSBAM_EINVAL
    eng.close()


@pytest.mark.parametrize("block,seed", [(5000, 3), (300, 4), (70, 5), (0xFF00, 6)])
def test_feed_records_crossing_blocks(pkg, tmp_path, monkeypatch, block, seed):
    """htsjdk-style layouts: BGZF blocks cut regardless of records (down to 70-byte blocks, so a record
    spans many blocks and even its length word is split).  The device stitches the chain from per-block
    pieces; with tiny super-batches the partial record at the end of one is carried into the next.
    Tables and status tallies must equal the host path's."""
    contigs, refs, recs = tl.fuzz_dataset(60 + seed, 2500)
    bam = tmp_path / "ragged.bam"
    tl.write_bam(bam, refs, recs, level=1, rng=np.random.default_rng(seed), block=block)
    raw = bam.read_bytes()
    hb = _bam_header_bytes(raw)
    rec_bytes = tl.raw_records(refs, recs)
    pss, kmer = dict(region_len=15 if seed % 2 else 25), dict(klen=5)
    eng = pkg.Engine(pss=pss, kmer=kmer)
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references([n for n, _ in refs])
    eng.submit(rec_bytes)
    want = eng.finish()
    eng.close()
    for super_bytes, max_batch in ((None, 1 << 30), ("1048576", 40000), ("1048576", 1 << 30)):
        if super_bytes:
            monkeypatch.setenv("PSSBAM_FEED_SUPER_BYTES", super_bytes)   # ~10 super-batches: tails are carried
        else:
            monkeypatch.delenv("PSSBAM_FEED_SUPER_BYTES", raising=False)
        got, st = _feed_tables(pkg, contigs, refs, raw, hb, pss, kmer, max_batch)
        assert st["flags"] == 0, (block, super_bytes, st)
        assert np.array_equal(got.fwd, want.fwd) and np.array_equal(got.rev, want.rev), (block, super_bytes)
        assert np.array_equal(got.k5, want.k5) and np.array_equal(got.k3, want.k3)
        a, b = dict(got.stats), dict(want.stats)
        a.pop("slow_path"), b.pop("slow_path")
        assert a == b, (block, super_bytes, got.stats, want.stats)


def _bgzf_block_starts(raw):
    """file offsets of the BGZF blocks of `raw` (BSIZE of the BC subfield, SAM spec 4.1)"""
    starts, p = [], 0
    while p < len(raw):
        starts.append(p)
        p += int.from_bytes(raw[p + 16:p + 18], "little") + 1
    assert p == len(raw)
    return starts


@pytest.mark.parametrize("block,seed,n_eng,per_run", [(300, 4, 2, 37), (70, 5, 3, 90), (0xFF00, 6, 2, 1), (5000, 3, 3, 2)])
def test_feed_handoff_between_engines(pkg, tmp_path, monkeypatch, block, seed, n_eng, per_run):
    """One BAM in htsjdk's layout dealt to several engines in alternating runs of blocks (what the front end does with
    n GPUs): the record a run ends in is handed to the engine that gets the next run (pssbam_engine_feed_handoff) and
    completed there.  Summed tables and status tallies equal the one-engine host path's; no flag is raised; a stream
    cut inside a record is diagnosed by the engine that holds its last run."""
    contigs, refs, recs = tl.fuzz_dataset(60 + seed, 2500)
    bam = tmp_path / "ragged.bam"
    tl.write_bam(bam, refs, recs, level=1, rng=np.random.default_rng(seed), block=block)
    raw = bam.read_bytes()
    hb = _bam_header_bytes(raw)
    pss, kmer = dict(region_len=15 if seed % 2 else 25), dict(klen=5)
    eng = pkg.Engine(pss=pss, kmer=kmer)
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references([n for n, _ in refs])
    eng.submit(tl.raw_records(refs, recs))
    want = eng.finish()
    eng.close()
    monkeypatch.setenv("PSSBAM_FEED_SUPER_BYTES", "1048576")

    def deal(data, header_bytes):
        starts = _bgzf_block_starts(data) + [len(data)]
        engs = []
        for _ in range(n_eng):
            e = pkg.Engine(pss=pss, kmer=kmer)
            e.set_genome_arrays(tl.loaded_contigs(contigs))
            e.set_references([n for n, _ in refs])
            engs.append(e)
        # (the first run takes every block the BAM header reaches into, and a few more)
        ends = np.cumsum([int.from_bytes(data[starts[i + 1] - 4:starts[i + 1]], "little") for i in range(len(starts) - 1)])   # ISIZE
        first = int(np.searchsorted(ends, header_bytes, side="right"))
        cuts = [0, first + per_run] + list(range(first + 2 * per_run, len(starts) - 1, per_run)) + [len(starts) - 1]
        cuts = sorted(set(c for c in cuts if c <= len(starts) - 1))
        prev = None
        for r, (b0, b1) in enumerate(zip(cuts[:-1], cuts[1:])):
            e = engs[r % n_eng]
            if prev is not None:
                prev.feed_handoff(e)
            e.submit_bgzf(np.frombuffer(data[starts[b0]:starts[b1]], dtype=np.uint8), header_bytes=header_bytes if r == 0 else 0)
            prev = e
        flags = [e.feed_status()["flags"] for e in engs]
        tabs = [e.finish() for e in engs]
        for e in engs:
            e.close()
        return flags, tabs, (len(cuts) - 2) % n_eng

    flags, tabs, _ = deal(raw, hb)
    assert flags == [0] * n_eng, flags
    assert np.array_equal(sum(t.fwd for t in tabs), want.fwd) and np.array_equal(sum(t.rev for t in tabs), want.rev)
    assert np.array_equal(sum(t.k5 for t in tabs), want.k5) and np.array_equal(sum(t.k3 for t in tabs), want.k3)
    got = {}
    for t in tabs:
        for k, v in dict(t.stats).items():
            got[k] = got.get(k, 0) + v
    b = dict(want.stats)
    got.pop("slow_path"), b.pop("slow_path")
    assert got == b, (got, b)
    assert min(dict(t.stats)["records"] for t in tabs) > 0          # every engine did tally a share
    # the same stream cut in the middle of a record: only the engine with the last run says so
    data = tl.bgzf_inflate(raw)
    cut = data[:int(len(data) * 0.7) + 11]
    reblocked = b"".join(tl.bgzf_block(cut[i:i + block], 1) for i in range(0, len(cut), block)) + tl.BGZF_EOF
    flags, _, last = deal(reblocked, hb)
    assert flags[last] & 8 and all(f == 0 for i, f in enumerate(flags) if i != last), (flags, last)


@pytest.mark.parametrize("trailer", [False, True])
def test_feed_repairs_a_false_record_start(pkg, tmp_path, monkeypatch, trailer):
    """Bytes INSIDE a record that read like a chain of records (here: a B:C aux array holding the raw bytes of twelve
    other records) and a BGZF block that begins exactly there: the block's own guess at its first record start is wrong,
    its chain does not link up with the one coming from the left.  Round 2 gave such a file up (PSSBAM_FEED_RAGGED ->
    host reader); now bgzf_chain_repair walks the blocks in doubt from the left and the feed carries on -- same tables
    as the host path, the fake records not tallied.  PSSBAM_FEED_REPAIR=0 shows the input does break the links."""
    contigs, refs, recs = tl.fuzz_dataset(77, 1200, extras=False)
    idx = {n: i for i, (n, _) in enumerate(refs)}
    rng = np.random.default_rng(5)
    hosts = sorted(int(x) for x in rng.choice(np.arange(50, 1150), 6, replace=False))
    fake_at = {}
    for h in hosts:
        fake = b"".join(tl.bam_record(recs[int(k)], idx) for k in rng.choice(len(recs), 12, replace=False))
        tags = [("XB", "B", ("C", list(fake)))]
        if trailer:
            tags.append(("XZ", "Z", "behind the array"))
        recs[h] = tl.Rec(**{**recs[h].__dict__, "tags": list(recs[h].tags) + tags})
        fake_at[h] = fake
    head = tl.bam_bytes(refs, [])
    body = [tl.bam_record(r, idx) for r in recs]
    raw = head + b"".join(body)
    # block cuts: every 700 bytes, plus one exactly where each fake chain begins
    starts = np.cumsum([len(head)] + [len(b) for b in body])
    cuts = set(range(0, len(raw), 700))
    for h in hosts:
        at = raw.index(fake_at[h], int(starts[h]))
        assert at < starts[h + 1]
        cuts.add(at)
    cuts = sorted(cuts) + [len(raw)]
    bgzf = b"".join(tl.bgzf_block(raw[a:b], 6) for a, b in zip(cuts[:-1], cuts[1:])) + tl.BGZF_EOF
    pss, kmer = dict(region_len=12), dict(klen=4)
    eng = pkg.Engine(pss=pss, kmer=kmer)
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references([n for n, _ in refs])
    eng.submit(tl.raw_records(refs, recs))
    want = eng.finish()
    eng.close()
    for super_bytes in (None, "65536"):
        if super_bytes:
            monkeypatch.setenv("PSSBAM_FEED_SUPER_BYTES", super_bytes)
        monkeypatch.setenv("PSSBAM_FEED_REPAIR", "0")
        _, st = _feed_tables(pkg, contigs, refs, bgzf, len(head), pss, kmer, 30000)
        assert st["flags"] & 2, st                                  # the links ARE broken
        monkeypatch.delenv("PSSBAM_FEED_REPAIR")
        got, st = _feed_tables(pkg, contigs, refs, bgzf, len(head), pss, kmer, 30000)
        assert st["flags"] == 0, (super_bytes, st)
        assert np.array_equal(got.fwd, want.fwd) and np.array_equal(got.rev, want.rev)
        assert np.array_equal(got.k5, want.k5) and np.array_equal(got.k3, want.k3)
        a, b = dict(got.stats), dict(want.stats)
        a.pop("slow_path"), b.pop("slow_path")
        assert a == b, (got.stats, want.stats)


@pytest.mark.parametrize("inflate_streams", ["1", "2"])
def test_feed_flags_damage_and_truncation(pkg, tmp_path, monkeypatch, inflate_streams):
    """a damaged block raises PSSBAM_FEED_BAD_BLOCK; a stream that ends inside a record raises
    PSSBAM_FEED_TRUNCATED -- also with the inflate launches on streams of their own and the 16 KiB-table form of the CRC
    kernel behind them (PSSBAM_FEED_INFLATE_STREAMS=2: a switch that is off by default and must not rot)"""
    monkeypatch.setenv("PSSBAM_FEED_INFLATE_STREAMS", inflate_streams)
    monkeypatch.setenv("PSSBAM_FEED_SUPER_BYTES", str(1 << 20))   # several super-batches: the streams do take turns
    contigs, refs, recs = tl.fuzz_dataset(31, 3000)
    bam = tmp_path / "a.bam"
    hb = tl.write_bam_aligned(bam, refs, recs, level=1, rng=np.random.default_rng(3))
    good = bam.read_bytes()
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x10
    _, st = _feed_tables(pkg, contigs, refs, bytes(bad), hb, dict(region_len=15))
    assert st["flags"] & 1
    # cut the inflated stream in the middle of a record: re-block the first 60 % of it
    data = tl.bgzf_inflate(good)
    cut = data[:int(len(data) * 0.6)]
    blocks = b"".join(tl.bgzf_block(cut[i:i + 0xFF00], 1) for i in range(0, len(cut), 0xFF00)) + tl.BGZF_EOF
    _, st = _feed_tables(pkg, contigs, refs, blocks, hb, dict(region_len=15))
    assert st["flags"] & 8 and not st["flags"] & 1


def test_cli_device_feed_and_fallback(pkg, oracle, tmp_path):
    """bin/pss-bam / bin/fragkon with the inflate on the device: htslib-layout BAMs and BAMs whose
    records cross BGZF blocks are fed compressed (many small chunks and submits, two engines),
    PSSBAM_DEVICE_INFLATE=0 keeps the host path -- identical tables every way"""
    import os
    import re
    import subprocess
    from pss_bam_amd import synth
    d = synth.config("C4", scale_genome=0.002, n_reads=400_000)
    region_len = d.pop("region_len")
    cfg = synth.make_cfg(**d)
    fa, sam = tmp_path / "g.fa", tmp_path / "a.sam"
    synth.fasta_host(cfg, fa)
    synth.sam_host(cfg, 0, 400_000, sam)
    g = oracle.load_genome(fa)
    po, ko = tl.PssOpts(region_len=region_len, min_mq=20), tl.FkOpts(klen=5)
    wf, wr, st = oracle.pss(g, sam, po)
    w5, w3, _ = oracle.fragkon(g, sam, ko)
    oracle.free_genome(g)
    aligned, ragged = tmp_path / "aligned.bam", tmp_path / "ragged.bam"
    synth.bam_file_host(cfg, 0, 400_000, aligned, level=1, threads=4)
    synth.bam_file_host(cfg, 0, 400_000, ragged, level=6, threads=4, ragged=True)
    b = pkg.PKG_DIR / "bin"

    def run_pss(bam, extra):
        env = {**os.environ, "PSSBAM_STATS": "1", **extra}
        pr = subprocess.run([str(b / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / "o")] + po.argv(),
                            capture_output=True, text=True, env=env)
        assert pr.returncode == 0, pr.stderr[-3000:]
        gf, gr = tl.parse_counts_text((tmp_path / "o.pss.counts.txt").read_text())
        assert np.array_equal(gf, wf) and np.array_equal(gr, wr), extra
        assert f"[pssbam] records=400000" in pr.stderr and f"[pssbam] pss_ok={st[tl.ST_OK]}" in pr.stderr
        return pr.stderr

    err = run_pss(aligned, {})
    m = re.search(r"device feed: (\d+) submits", err)
    assert m and "host reader" not in err, err[-1500:]
    # by default the feed runs on a helper thread WHILE the FASTA loads (tallies put off until the genome is
    # posted); PSSBAM_NO_EARLY_FEED keeps the serial order of the reference -- same tables either way
    assert "early feed (helper thread" in err, err[-1500:]
    err = run_pss(aligned, {"PSSBAM_NO_EARLY_FEED": "1"})
    assert "early feed (helper thread" not in err and re.search(r"device feed: (\d+) submits", err)
    err = run_pss(aligned, {"PSSBAM_FEED_MAX_SLOTS": "1", "PSSBAM_FEED_SUPER_BYTES": str(2 << 20), "PSSBAM_FEED_BATCH_BYTES": str(1 << 20),
                            "PSSBAM_CHUNK_BYTES": str(1 << 20)})       # the ring fills before the genome comes: the feed waits for it
    assert "early feed (helper thread" in err
    err = run_pss(aligned, {"PSSBAM_CHUNK_BYTES": str(1 << 20), "PSSBAM_FEED_BATCH_BYTES": str(3 << 20), "PSSBAM_NGPU": "2",
                            "PSSBAM_OVERSUBSCRIBE": "1", "PSSBAM_LOADER_THREADS": "3", "PSSBAM_RUN_BATCHES": "2"})
    m = re.search(r"device feed: (\d+) submits", err)
    assert m and int(m.group(1)) >= 10 and "gpus=2" in err
    err = run_pss(ragged, {})                                   # records cross BGZF blocks: still fed compressed
    assert re.search(r"device feed: (\d+) submits", err) and "host reader" not in err
    err = run_pss(aligned, {"PSSBAM_DEVICE_INFLATE": "0"})
    assert "device feed" not in err
    err = run_pss(ragged, {"PSSBAM_FEED_INFLATE_STREAMS": "2", "PSSBAM_FEED_SUPER_BYTES": str(4 << 20), "PSSBAM_FEED_BATCH_BYTES": str(1 << 20),
                           "PSSBAM_CHUNK_BYTES": str(1 << 20)})   # the off-by-default two-stream form of the feed, many super-batches
    assert re.search(r"device feed: (\d+) submits", err) and "host reader" not in err
    # two engines are dealt alternating runs of the file: the record a run ends in is handed to the engine that
    # gets the next run (pssbam_engine_feed_handoff) -- a file in htsjdk's layout stays on the device feed
    err = run_pss(ragged, {"PSSBAM_CHUNK_BYTES": str(1 << 20), "PSSBAM_NGPU": "2", "PSSBAM_OVERSUBSCRIBE": "1",
                           "PSSBAM_FEED_BATCH_BYTES": str(3 << 20), "PSSBAM_RUN_BATCHES": "2"})
    assert "falling back to the host reader" not in err and "host reader" not in err and "gpus=2" in err
    m = re.search(r"submits per engine: (\d+) (\d+)", err)
    assert m and int(m.group(1)) > 0 and int(m.group(2)) > 0, err[-1500:]
    err = run_pss(ragged, {"PSSBAM_CHUNK_BYTES": str(1 << 20), "PSSBAM_NGPU": "3", "PSSBAM_OVERSUBSCRIBE": "1", "PSSBAM_NO_EARLY_FEED": "1",
                           "PSSBAM_FEED_BATCH_BYTES": str(1 << 20), "PSSBAM_RUN_BATCHES": "1", "PSSBAM_FEED_SUPER_BYTES": str(2 << 20)})
    assert "host reader" not in err and "gpus=3" in err
    pr = subprocess.run([str(b / "fragkon"), "-F", str(fa), "-B", str(aligned)] + ko.argv(), capture_output=True, text=True)
    assert pr.returncode == 0, pr.stderr
    g5, g3 = tl.parse_fragkon_text(pr.stdout)
    assert np.array_equal(g5, w5) and np.array_equal(g3, w3)
    # a damaged block is a diagnosed failure, not a wrong table
    raw = bytearray(aligned.read_bytes())
    raw[len(raw) // 2] ^= 0x20
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(raw))
    pr = subprocess.run([str(b / "pss-bam"), "-F", str(fa), "-B", str(bad), "-o", str(tmp_path / "x")] + po.argv(),
                        capture_output=True, text=True)
    assert pr.returncode != 0 and "Error" in pr.stderr


def test_cli_device_feed_edge_files(pkg, tmp_path):
    """header-only BAM -> all-zero tables; a BAM cut in the middle of a BGZF block, or in the middle of a
    record, -> a diagnosed failure (not a table)"""
    import os
    import subprocess
    contigs, refs, recs = tl.fuzz_dataset(77, 1500)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    b = pkg.PKG_DIR / "bin"

    def run(bam, prefix):
        return subprocess.run([str(b / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / prefix), "-r", "10"],
                              capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1"})

    empty = tmp_path / "empty.bam"
    tl.write_bam_aligned(empty, refs, [], level=6)
    pr = run(empty, "e")
    assert pr.returncode == 0, pr.stderr
    f, r = tl.parse_counts_text((tmp_path / "e.pss.counts.txt").read_text())
    assert f.sum() == 0 and r.sum() == 0 and "[pssbam] records=0" in pr.stderr
    full = tmp_path / "full.bam"
    tl.write_bam_aligned(full, refs, recs, level=6)
    raw = full.read_bytes()
    cut_block = tmp_path / "cut_block.bam"
    cut_block.write_bytes(raw[:len(raw) // 2])                      # ends inside a BGZF block
    pr = run(cut_block, "c1")
    assert pr.returncode != 0 and "Error" in pr.stderr
    data = tl.bgzf_inflate(raw)
    part = data[:int(len(data) * 0.7)]                               # ends inside an alignment record, blocks intact
    cut_rec = tmp_path / "cut_rec.bam"
    cut_rec.write_bytes(b"".join(tl.bgzf_block(part[i:i + 0xFF00], 6) for i in range(0, len(part), 0xFF00)) + tl.BGZF_EOF)
    pr = run(cut_rec, "c2")
    assert pr.returncode != 0 and "truncated" in pr.stderr.lower(), pr.stderr[-1500:]


def test_cli_damaged_record_stream_same_outcome_in_both_feeds(pkg, tmp_path):
    """random overwrites in the RECORD stream, BGZF-compressed afterwards (every CRC is right): the device-side
    record chain meets broken block_size fields and cut-off records.  The command may fail with a diagnosis or
    fall back to the host reader -- the outcome must be the host reader's (same tables, or a failure in both),
    never a signal (tools/feed_soak.py is the long version)"""
    import os
    import subprocess
    contigs, refs, recs = tl.fuzz_dataset(123, 3000)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    good = tmp_path / "good.bam"
    tl.write_bam_aligned(good, refs, recs, level=6)
    data = bytearray(tl.bgzf_inflate(good.read_bytes()))
    header_end = _bam_header_bytes(good.read_bytes())
    rng = np.random.default_rng(9)
    b = pkg.PKG_DIR / "bin" / "pss-bam"
    for f in range(6):
        d = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            at = int(rng.integers(header_end, len(d) - 8))
            m = int(rng.integers(1, 6))
            d[at:at + m] = bytes(rng.integers(0, 256, m, dtype=np.uint8))
        blk = [300, 5000, 0xFF00][f % 3]
        bam = tmp_path / "bad.bam"
        bam.write_bytes(b"".join(tl.bgzf_block(bytes(d[i:i + blk]), 6) for i in range(0, len(d), blk)) + tl.BGZF_EOF)
        res = []
        for tag, env in (("d", {}), ("h", {"PSSBAM_DEVICE_INFLATE": "0"})):
            pr = subprocess.run([str(b), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / tag), "-r", "10"],
                                capture_output=True, text=True, env={**os.environ, **env}, timeout=120)
            assert pr.returncode >= 0, (f, tag, pr.returncode, pr.stderr[-400:])
            res.append(pr.returncode)
        assert (res[0] == 0) == (res[1] == 0), (f, res)
        if res[0] == 0:
            assert (tmp_path / "d.pss.counts.txt").read_text().split("\n", 6)[-1] == (tmp_path / "h.pss.counts.txt").read_text().split("\n", 6)[-1], f


def test_feed_ahead_of_the_genome(pkg, tmp_path, monkeypatch):
    """pssbam_engine_feed_open: the compressed file goes in BEFORE set_genome / set_references -- inflate, CRC and
    record index run at once, the tally launches follow when the genome arrives.  Same tables as the serial order;
    with a ring that may not grow (PSSBAM_FEED_MAX_SLOTS) the engine answers PSSBAM_EBUSY, takes nothing of the
    chunk, and carries on once the genome is set."""
    contigs, refs, recs = tl.fuzz_dataset(515, 24000, with_rg=True)                      # ~9 MB of records
    bam = tmp_path / "a.bam"
    tl.write_bam(bam, refs, recs, level=6, rng=np.random.default_rng(8), block=4000)    # records cross blocks
    raw = np.frombuffer(bam.read_bytes(), dtype=np.uint8)
    hb = _bam_header_bytes(bam.read_bytes())
    pss, kmer = dict(region_len=20, min_mq=5), dict(klen=3)
    eng = pkg.Engine(pss=pss, kmer=kmer, read_group="grpB")
    eng.set_genome_arrays(tl.loaded_contigs(contigs))
    eng.set_references([n for n, _ in refs])
    eng.submit(tl.raw_records(refs, recs))
    want = eng.finish()
    eng.close()
    monkeypatch.setenv("PSSBAM_FEED_SUPER_BYTES", str(1 << 20))      # ~9 super-batches
    for max_slots, batch in ((None, 1 << 30), ("2", 60000), ("1", 50000), ("40", 200000)):   # (ONE submit is never cut short: it may pass the cap)
        if max_slots:
            monkeypatch.setenv("PSSBAM_FEED_MAX_SLOTS", max_slots)
        else:
            monkeypatch.delenv("PSSBAM_FEED_MAX_SLOTS", raising=False)
        eng = pkg.Engine(pss=pss, kmer=kmer, read_group="grpB")
        with pytest.raises(pkg.PssbamError):                          # neither genome nor feed_open: refused
            eng.submit_bgzf(raw, header_bytes=hb)
        eng.feed_open(len(refs))
        busy = []

        def set_genome():
            busy.append(1)
            eng.set_genome_arrays(tl.loaded_contigs(contigs))
            eng.set_references([n for n, _ in refs])
        eng.submit_bgzf(raw, header_bytes=hb, max_batch_inflated=batch, on_busy=set_genome)
        ebusy = bool(busy)
        assert ebusy == (max_slots in ("1", "2")), (max_slots, busy)  # a ring that may not grow fills up; one that may does not
        if not ebusy:
            with pytest.raises(pkg.PssbamError):                      # fed, but the genome never came
                eng.sync()
            set_genome()
        st = eng.feed_status()
        got = eng.finish()
        eng.close()
        assert st["flags"] == 0, (max_slots, st)
        assert np.array_equal(got.fwd, want.fwd) and np.array_equal(got.rev, want.rev), max_slots
        assert np.array_equal(got.k5, want.k5) and np.array_equal(got.k3, want.k3), max_slots
        a, b = dict(got.stats), dict(want.stats)
        a.pop("slow_path"), b.pop("slow_path")
        assert a == b, (max_slots, a, b)


@pytest.mark.parametrize("seed", range(4))
def test_submit_bgzf_matches_the_oracle(pkg, oracle, tmp_path, seed):
    """the compressed feed against the ORACLE (not against the engine's own host path): random records incl. aux
    fields of every type, random options, both BGZF layouts, with and without -R"""
    with_rg = seed % 2 == 1
    contigs, refs, recs = tl.fuzz_dataset(8800 + seed, 3000, with_rg=with_rg, extras=True)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    bam = tmp_path / "a.bam"
    if seed < 2:
        hb = tl.write_bam_aligned(bam, refs, recs, level=1 + seed * 5, rng=np.random.default_rng(seed))
    else:
        tl.write_bam(bam, refs, recs, level=6, rng=np.random.default_rng(seed), block=[900, 30000][seed - 2])
        hb = _bam_header_bytes(bam.read_bytes())
    raw = np.frombuffer(bam.read_bytes(), dtype=np.uint8)
    g = oracle.load_genome(fa)
    rng = np.random.default_rng(100 + seed)
    try:
        for trial in range(3):
            po, ko = tl.random_pss_opts(rng), tl.random_fk_opts(rng)
            rg = [None, "grpA", "grpB"][trial] if with_rg else None
            keep = recs if rg is None else [r for r in recs if ("RG", "Z", rg) in r.tags]
            sam = tmp_path / f"t{trial}.sam"
            tl.write_sam(sam, refs, keep)
            wf, wr, st = oracle.pss(g, sam, po)
            w5, w3, stk = oracle.fragkon(g, sam, ko)
            eng = pkg.Engine(pss=dict(region_len=po.region_len, min_read_len=po.min_read_len, max_read_len=po.max_read_len,
                                      min_mq=po.min_mq, up_ctx=po.up_ctx, down_ctx=po.down_ctx, merged_only=po.merged_only),
                             kmer=dict(klen=ko.klen, min_mq=ko.min_mq, min_read_len=ko.min_read_len, max_read_len=ko.max_read_len,
                                       merged_only=ko.merged_only), read_group=rg)
            if trial == 1:
                eng.feed_open(len(refs))                 # ahead of the genome
                eng.submit_bgzf(raw, header_bytes=hb, max_batch_inflated=300000)
            eng.set_genome_arrays(tl.loaded_contigs(contigs))
            eng.set_references([n for n, _ in refs])
            if trial != 1:
                eng.submit_bgzf(raw, header_bytes=hb, max_batch_inflated=[1 << 30, 0, 90000][trial])
            assert eng.feed_status()["flags"] == 0
            got = eng.finish()
            eng.close()
            assert np.array_equal(got.fwd, wf) and np.array_equal(got.rev, wr), (seed, trial, po)
            assert np.array_equal(got.k5, w5.astype(np.uint64)) and np.array_equal(got.k3, w3.astype(np.uint64)), (seed, trial, ko)
            assert got.stats["records"] == len(recs) and got.stats["rg_dropped"] == len(recs) - len(keep)
            assert got.stats["pss_ok"] == st[tl.ST_OK] and got.stats["pss_filtered"] == st[tl.ST_FILTERED]
            assert got.stats["no_contig"] == st[tl.ST_NO_CONTIG] and got.stats["parse_skip"] == st[tl.ST_PARSE_SKIP]
            assert got.stats["kmer_ok"] == stk[tl.ST_OK] and got.stats["kmer_fail"] == stk[tl.ST_KMER_FAIL]
    finally:
        oracle.free_genome(g)


def test_empty_block_behind_the_header_and_a_bad_first_record(pkg, tmp_path):
    """an empty BGZF block right behind the header block leaves block 0 of the feed without a record: the chain's
    anchor must still be checked against the first candidate a later block finds.  With a first record whose refID
    equals n_ref (implausible) the device feed may not drop it silently: same outcome as the host reader"""
    import os
    import subprocess
    contigs, refs, recs = tl.fuzz_dataset(4321, 1200)
    fa = tmp_path / "g.fa"
    tl.write_fasta(fa, contigs)
    idx = {n: i for i, (n, _) in enumerate(refs)}
    head = tl.bam_bytes(refs, [])
    body = [tl.bam_record(r, idx) for r in recs]
    bad_first = bytearray(body[0])
    bad_first[4:8] = struct.pack("<i", len(refs))                    # refID == n_ref
    b = pkg.PKG_DIR / "bin" / "pss-bam"
    outcomes = {}
    for name, first in (("good", body[0]), ("bad", bytes(bad_first))):
        data = first + b"".join(body[1:])
        bam = tmp_path / f"{name}.bam"
        bam.write_bytes(tl.bgzf_block(head, 6) + tl.bgzf_block(b"", 6)
                        + b"".join(tl.bgzf_block(data[i:i + 0xFF00], 6) for i in range(0, len(data), 0xFF00)) + tl.BGZF_EOF)
        res = []
        for tag, env in (("d", {}), ("h", {"PSSBAM_DEVICE_INFLATE": "0"})):
            pr = subprocess.run([str(b), "-F", str(fa), "-B", str(bam), "-o", str(tmp_path / (name + tag)), "-r", "10"],
                                capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1", **env}, timeout=120)
            assert pr.returncode >= 0, (name, tag, pr.stderr[-500:])
            res.append((pr.returncode == 0, (tmp_path / f"{name}{tag}.pss.counts.txt").read_text().split("\n", 6)[-1] if pr.returncode == 0 else None,
                        re_records(pr.stderr)))
        assert res[0] == res[1], (name, res[0][0], res[1][0], res[0][2], res[1][2])
        outcomes[name] = res[0]
    assert outcomes["good"][0] and outcomes["good"][2] == len(recs)


def re_records(stderr: str):
    import re
    m = re.search(r"\[pssbam\] records=(\d+)", stderr)
    return int(m.group(1)) if m else None
