"""CPU tests of the host C side (pss-bam_amd/host): reference-API-compatible modules, the
BGZF/BAM reader, the report writers and the front ends' error paths.  The oracle and the
golden vectors are the checkers."""
import ctypes as C
import json
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

import __graft_entry__ as ge
import pssbam_testlib as tl

GOLD = Path(__file__).resolve().parent / "golden"
MANIFEST = json.loads((GOLD / "manifest.json").read_text())


@pytest.fixture(scope="module")
def host():
    ge.build()
    pkg = ge.load_pkg()
    L = C.CDLL(str(pkg.LIB_HOST))
    L.init_genome.restype = C.c_void_p
    L.init_genome.argtypes = [C.c_char_p]
    L.destroy_genome.argtypes = [C.c_void_p]
    L.find_seq.restype = C.c_void_p
    L.find_seq.argtypes = [C.c_void_p, C.c_char_p]
    L.line2saml.argtypes = [C.c_char_p, C.c_void_p]
    L.aln_seq_len.argtypes = [C.c_char_p]
    L.init_KSP.restype = C.c_void_p
    L.add_to_ksp.argtypes = [C.c_char_p, C.c_void_p]
    L.kmer2count.argtypes = [C.c_char_p, C.c_void_p]
    L.kmer2count.restype = C.c_uint
    L.kmer2inx.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.destroy_KSP.argtypes = [C.c_void_p]
    L.pss_sub_rates.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.pss_write_counts.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    L.pss_write_rates.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    return L, pkg


class _Seq(C.Structure):
    _fields_ = [("id", C.c_char * 512), ("seq", C.c_void_p), ("len", C.c_size_t)]


class _Genome(C.Structure):
    _fields_ = [("seqs", C.POINTER(C.POINTER(_Seq))), ("dummy", C.c_void_p), ("n_seqs", C.c_size_t)]


def _genome_contents(L, path):
    g = L.init_genome(str(path).encode())
    assert g, f"init_genome failed for {path}"
    G = C.cast(g, C.POINTER(_Genome)).contents
    out = []
    for i in range(G.n_seqs):
        s = G.seqs[i].contents
        out.append((s.id.decode(), C.string_at(s.seq, s.len), s.len))
    return g, out


@pytest.mark.parametrize("gz", [False, True])
def test_genome_loader_matches_reference_semantics(host, tmp_path, gz):
    """upper-casing, whitespace stripping, id = first word, sort by id, '>' splits anywhere"""
    L, _ = host
    rng = np.random.default_rng(11)
    contigs = [("zeta", tl.random_contig(rng, 70001)), ("alpha", tl.random_contig(rng, 333)), ("Mid.1", "acgtNNryACGT"),
               ("empty", "")]
    fa = tmp_path / ("g.fa.gz" if gz else "g.fa")
    tl.write_fasta(fa, contigs, width=61, gz=gz)
    g, got = _genome_contents(L, fa)
    want = sorted((cid, seq.upper().encode()) for cid, seq in contigs)
    assert [(a, b) for a, b, _ in got] == want
    assert all(n == len(b) for _, b, n in got)
    # find_seq: exact names only
    assert L.find_seq(g, b"alpha") and L.find_seq(g, b"zeta") and not L.find_seq(g, b"alph") and not L.find_seq(g, b"*")
    L.destroy_genome(g)


def test_genome_loader_odd_layouts(host, tmp_path, oracle):
    L, _ = host
    fa = tmp_path / "odd.fa"
    # CRLF line ends, blank lines, tabs inside the body, no final newline, '>' glued to a body line
    fa.write_bytes(b">c1 first\r\nACgt\r\n\r\n ac\tgt \r\n>c2\nTTTT>c3 x\nGGGG\nCC")
    g, got = _genome_contents(L, fa)
    assert [(a, b) for a, b, _ in got] == [("c1", b"ACGTACGT"), ("c2", b"TTTT"), ("c3", b"GGGGCC")]
    L.destroy_genome(g)
    # malformed inputs are diagnosed (the reference has undefined behaviour on them)
    bad = tmp_path / "bad.fa"
    bad.write_bytes(b"ACGT\n>x\nAC\n")
    assert not L.init_genome(str(bad).encode())
    assert not L.init_genome(str(tmp_path / "missing.fa").encode())


def test_genome_loader_parallel_equals_serial(host, tmp_path, oracle):  # noqa: C901
    """files above 1 MiB take the multi-threaded path: same contigs as the one-thread parser and
    as the oracle's loader, for layouts that stress the piece stitching (a '>' glued to a body
    line, very long and very short lines, CRLF, empty contigs, a contig spanning many pieces)"""
    L, _ = host
    rng = np.random.default_rng(12)
    big = tl.random_contig(rng, 2_200_000)
    contigs = [("chrBig", big), ("empty1", ""), ("chrLongLine", tl.random_contig(rng, 400_000)),
               ("tiny", "acgtn"), ("chrCR", tl.random_contig(rng, 300_000)), ("last", tl.random_contig(rng, 1000))]
    fa = tmp_path / "par.fa"
    with open(fa, "wb") as fh:
        fh.write(b">chrBig some description > with a bracket\n")
        for i in range(0, len(big), 71):
            fh.write(big[i:i + 71].encode() + b"\n")
        fh.write(b">empty1\n")
        fh.write(b">chrLongLine\n" + contigs[2][1].encode() + b"\n")          # one 400 kb line
        fh.write(b">tiny\tdescr\nac\ngt\nn")                                  # next '>' glued to the body line
        fh.write(b">chrCR x\r\n")
        for i in range(0, 300_000, 50):
            fh.write(contigs[4][1][i:i + 50].encode() + b"\r\n")
        fh.write(b"\n\n>last\n" + contigs[5][1].encode())                      # no final newline
    assert fa.stat().st_size > (1 << 20)
    want = sorted((cid, seq.upper().encode()) for cid, seq in contigs)
    for threads in ("1", "3", "16"):
        os.environ["PSSBAM_FASTA_THREADS"] = threads
        try:
            g, got = _genome_contents(L, fa)
        finally:
            del os.environ["PSSBAM_FASTA_THREADS"]
        assert [(a, b) for a, b, _ in got] == want, threads
        L.destroy_genome(g)
    # the oracle's loader (restating fasta-genome-io.c) sees the same genome: k-mer census of both
    og = oracle.load_genome(fa)
    os.environ["PSSBAM_FASTA_THREADS"] = "16"
    try:
        g, got = _genome_contents(L, fa)
    finally:
        del os.environ["PSSBAM_FASTA_THREADS"]
    mine = oracle.genome_from_arrays([(a, np.frombuffer(b, dtype=np.uint8).copy()) for a, b, _ in got])
    assert np.array_equal(oracle.genome_kmer_count(og, 5), oracle.genome_kmer_count(mine, 5))
    L.destroy_genome(g)
    # malformed: a header line cut by the end of the file, bytes before the first '>'
    cut = tmp_path / "cut.fa"
    cut.write_bytes(fa.read_bytes() + b"\n>unterminated header")
    assert not L.init_genome(str(cut).encode())
    pre = tmp_path / "pre.fa"
    pre.write_bytes(b"ACGT\n" + fa.read_bytes())
    assert not L.init_genome(str(pre).encode())


def test_genome_loader_bgzf_fasta(host, tmp_path):
    """a .gz reference written by bgzip (BGZF blocks: what `samtools faidx` needs) is inflated by all threads at
    once and parsed like a plain file; plain gzip keeps going through gzread; a damaged block is diagnosed"""
    import gzip
    L, _ = host
    rng = np.random.default_rng(21)
    contigs = [("chr2", tl.random_contig(rng, 1_400_000)), ("chr1", tl.random_contig(rng, 900_000)), ("tiny", "acgtn"), ("e", "")]
    plain = tmp_path / "ref.fa"
    tl.write_fasta(plain, contigs, width=60)
    raw = plain.read_bytes()
    assert len(raw) > (2 << 20)
    want = sorted((cid, seq.upper().encode()) for cid, seq in contigs)
    bg = tmp_path / "ref.bgzf.fa.gz"
    bg.write_bytes(b"".join(tl.bgzf_block(raw[i:i + 0xFF00], 6) for i in range(0, len(raw), 0xFF00)) + tl.BGZF_EOF)
    gz = tmp_path / "ref.plain.fa.gz"
    gz.write_bytes(gzip.compress(raw, 1))
    gz2 = tmp_path / "ref.two_members.fa.gz"   # concatenated members, a header with a file name
    gz2.write_bytes(gzip.compress(raw[:1_000_001], 6) + b"\x1f\x8b\x08\x08\0\0\0\0\0\x03part2.fa\0" + gzip.compress(raw[1_000_001:], 9)[10:])
    gz3 = tmp_path / "ref.named.fa.gz"          # one member, FNAME set (what `gzip ref.fa` writes)
    with open(gz3, "wb") as fh, gzip.GzipFile(filename="ref.fa", mode="wb", fileobj=fh, compresslevel=6, mtime=0) as z:
        z.write(raw)
    assert gz3.read_bytes()[3] & 8
    for path in (bg, gz, gz2, gz3):
        for threads in ("16", "3", "1"):
            os.environ["PSSBAM_FASTA_THREADS"] = threads
            try:
                g, got = _genome_contents(L, path)
            finally:
                del os.environ["PSSBAM_FASTA_THREADS"]
            assert [(a, b) for a, b, _ in got] == want, (path.name, threads)
            L.destroy_genome(g)
    bad = bytearray(bg.read_bytes())
    bad[len(bad) // 2] ^= 0x10
    dmg = tmp_path / "damaged.fa.gz"
    dmg.write_bytes(bytes(bad))
    assert not L.init_genome(str(dmg).encode())


def test_genome_loader_equals_oracle_on_golden(host, oracle):
    L, _ = host
    for ds in MANIFEST["datasets"].values():
        g, got = _genome_contents(L, GOLD / ds["fasta"])
        txt = (GOLD / ds["fasta"]).read_text()
        want = sorted((blk.split("\n", 1)[0].split()[0], "".join(blk.split("\n")[1:]).upper().encode())
                      for blk in txt.split(">")[1:])
        assert [(a, b) for a, b, _ in got] == want
        L.destroy_genome(g)


SAML_SIZE = 20536
OFF = dict(qname=0, flag=2048, bits=2052, rname=2054, pos=4104, mapq=4112, cigar=4116, mrnm=6164, mpos=8212, isize=8216,
           seq_len=8220, seq=8224, qual=10272, tags=12320)


def _saml(L, line: str):
    buf = C.create_string_buffer(SAML_SIZE)
    rc = L.line2saml(line.encode(), buf)
    if rc:
        return rc, None
    raw = buf.raw
    cstr = lambda o: raw[o:raw.index(b"\0", o)].decode()  # noqa: E731
    u32 = lambda o: int.from_bytes(raw[o:o + 4], "little")  # noqa: E731
    return 0, dict(qname=cstr(OFF["qname"]), flag=u32(OFF["flag"]), bits=int.from_bytes(raw[OFF["bits"]:OFF["bits"] + 2], "little"),
                   rname=cstr(OFF["rname"]), pos=int.from_bytes(raw[OFF["pos"]:OFF["pos"] + 8], "little"),
                   mapq=u32(OFF["mapq"]), cigar=cstr(OFF["cigar"]), mrnm=cstr(OFF["mrnm"]), mpos=u32(OFF["mpos"]),
                   isize=int.from_bytes(raw[OFF["isize"]:OFF["isize"] + 4], "little", signed=True),
                   seq_len=u32(OFF["seq_len"]), seq=cstr(OFF["seq"]), qual=cstr(OFF["qual"]), tags=cstr(OFF["tags"]))


def test_line2saml_fields_and_quirks(host):
    L, _ = host
    rc, s = _saml(L, "r1\t83\tchr1\t100\t37\t5M\t=\t90\t-15\tACGTN\tIIIII\tNM:i:1\tRG:Z:x\n")
    assert rc == 0 and s["qname"] == "r1" and s["flag"] == 83 and s["rname"] == "chr1" and s["pos"] == 100
    assert s["mapq"] == 37 and s["cigar"] == "5M" and s["mrnm"] == "=" and s["mpos"] == 90 and s["isize"] == -15
    assert s["seq"] == "ACGTN" and s["qual"] == "IIIII" and s["seq_len"] == 5 and s["tags"] == "NM:i:1\tRG:Z:x\n"
    assert s["bits"] == 83 & 0xFFF                   # twelve flag bitfields = FLAG bits 0x1..0x800
    rc, s = _saml(L, "r2\t16\tchr1\t7\t0\t3M\t*\t0\t999\tACG\tIII\n")
    assert rc == 0 and s["isize"] == 3               # unpaired: isize := strlen(SEQ)  (sam-parse.c:66-68)
    assert _saml(L, "r3\t0\tchr1\t7\t0\t3M\t*\t0\t0\tACG\t*\n")[0] == 1       # SEQ/QUAL length mismatch
    assert _saml(L, "r4\t0\tchr1\t7\t0\t3M\t*\t0\t0\tACG\n")[0] == 1           # ten fields
    assert _saml(L, "@HD\tVN:1.6\n")[0] == 1
    assert _saml(L, "r5\tx\tchr1\t7\t0\t3M\t*\t0\t0\tACG\tIII\n")[0] == 1     # FLAG not a number
    rc, s = _saml(L, "r6 0 chr1 7 0 3M * 0 0x10 ACG III\n")                       # blanks separate too; %i reads hex
    assert rc == 0 and s["pos"] == 7 and s["isize"] == 3 and s["rname"] == "chr1"
    rc, s = _saml(L, "r7\t1\tchr1\t7\t0\t3M\t*\t0\t0x10\tACG\tIII\n")
    assert rc == 0 and s["isize"] == 16
    rc, s = _saml(L, "r8\t-1\tchr1\t7\t0\t3M\t*\t0\t0\tACG\tIII\n")              # %u wraps a negative
    assert rc == 0 and s["flag"] == 0xFFFFFFFF
    assert _saml(L, "r9\t0\t" + "c" * 3000 + "\t7\t0\t3M\t*\t0\t0\tACG\tIII\n")[0] == 1   # field > 2047: rejected
    assert L.aln_seq_len(b"10M2I5M3S") == 15 and L.aln_seq_len(b"*") == 0


def test_line2saml_agrees_with_oracle_parser_on_fuzz(host, oracle, tmp_path):
    """same accept/reject decision and same fields as the oracle's scanf restatement"""
    L, _ = host
    _, refs, recs = tl.fuzz_dataset(31, 1200, with_rg=True)
    lib = oracle.lib

    class Aln(C.Structure):
        _fields_ = [("rname", C.c_char_p), ("cigar", C.c_char_p), ("seq", C.c_char_p), ("flag", C.c_uint), ("mapq", C.c_uint),
                    ("pos", C.c_ulong), ("isize", C.c_int), ("seq_len", C.c_int)]
    lib.orc_parse_line.argtypes = [C.c_char_p, C.POINTER(Aln), C.c_char_p]
    rng = np.random.default_rng(5)
    n_ok = 0
    for r in recs:
        line = tl.sam_line(r)
        if rng.random() < 0.05:
            line = line.replace("\t", " ", int(rng.integers(1, 4)))     # blanks as separators
        if rng.random() < 0.03:
            line = "\t".join(line.split("\t")[:int(rng.integers(3, 11))]) + "\n"   # truncated line
        a = Aln()
        scratch = C.create_string_buffer(6 * (len(line) + 2))
        want_rc = lib.orc_parse_line(line.encode(), C.byref(a), scratch)
        rc, s = _saml(L, line)
        assert rc == want_rc, line
        if rc == 0:
            n_ok += 1
            assert (s["rname"], s["cigar"], s["seq"]) == (a.rname.decode(), a.cigar.decode(), a.seq.decode())
            assert (s["flag"], s["mapq"], s["pos"], s["isize"], s["seq_len"]) == (a.flag, a.mapq, a.pos, a.isize, a.seq_len)
    assert n_ok > 900


def test_kmer_table_api(host):
    L, _ = host
    for k in (3, 8, 11):
        ks = L.init_KSP(k)
        kmers = ["ACGTACGTACGT"[:k], "TTTTTTTTTTTT"[:k], "acgtacgtacgt"[:k]]
        for km in kmers:
            assert L.add_to_ksp(km.encode(), ks) == 0
        assert L.add_to_ksp(("ACGTNCGTACGT"[:k]).encode(), ks) == (-1 if k > 4 else 0)
        assert L.kmer2count(kmers[0].encode(), ks) == (2 if k > 4 else 3)          # case-folded on add
        assert L.kmer2count(kmers[1].encode(), ks) == 1 and L.kmer2count(("G" * k).encode(), ks) == 0
        L.destroy_KSP(ks)
    inx = C.c_size_t()
    assert L.kmer2inx(b"ACGT", 4, C.byref(inx)) == 1 and inx.value == 0b00011011
    assert L.kmer2inx(b"ACNT", 4, C.byref(inx)) == 0


@pytest.mark.parametrize("case", [c for c in MANIFEST["cases"] if c["tool"] == "pss-bam"], ids=lambda c: c["prefix"])
def test_report_writers_byte_exact(host, case, tmp_path):
    """tables parsed from the reference's counts file -> our writers -> identical files"""
    L, _ = host
    ds = MANIFEST["datasets"][case["dataset"]]
    want_counts = (GOLD / case["counts"]).read_text()
    fwd, rev = tl.parse_counts_text(want_counts)
    n = fwd.shape[0] - 2
    fr, rr = np.zeros((max(n, 1), 12)), np.zeros((max(n, 1), 12))
    L.pss_sub_rates(n, fwd.ctypes.data, fr.ctypes.data)
    L.pss_sub_rates(n, rev.ctypes.data, rr.ctypes.data)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        L.pss_write_counts(ds["fasta"].encode(), ds["sam"].encode(), case["prefix"].encode(), n, fwd.ctypes.data, rev.ctypes.data)
        L.pss_write_rates(ds["fasta"].encode(), ds["sam"].encode(), case["prefix"].encode(), n, fr.ctypes.data, rr.ctypes.data)
        assert Path(case["counts"]).read_text() == want_counts
        assert Path(case["rates"]).read_text() == (GOLD / case["rates"]).read_text()
    finally:
        os.chdir(cwd)


def test_bam_reader_roundtrip_through_bam2sam(host, tmp_path):
    """BGZF blocks of random size (records and even length words straddle blocks), several
    inflate threads, small batch buffer -> text identical to the independent SAM writer"""
    _, pkg = host
    exe = pkg.PKG_DIR / "bin" / "bam2sam"
    for seed, with_rg in ((1, False), (2, True)):
        _, refs, recs = tl.fuzz_dataset(500 + seed, 3000, with_rg=with_rg)
        bam = tmp_path / f"x{seed}.bam"
        tl.write_bam(bam, refs, recs, level=1, rng=np.random.default_rng(seed), block=4000)
        out = subprocess.run([str(exe), str(bam)], capture_output=True, text=True, check=True).stdout
        # tiny batch buffers: many batches, records carried across them, background prefetch busy
        out_small = subprocess.run([str(exe), str(bam)], capture_output=True, text=True, check=True,
                                   env={**os.environ, "PSSBAM_BATCH_BYTES": "262144"}).stdout
        assert out_small == out
        want = "".join(tl.sam_line(r) for r in recs)
        # BAM cannot tell "RNAME not in header" from '*': the writer maps both to refID -1
        want = want.replace("\tchrNotInHeader\t", "\t*\t")
        assert out == want
        if with_rg:
            out = subprocess.run([str(exe), "-r", "grpA", str(bam)], capture_output=True, text=True, check=True).stdout
            assert out == "".join(tl.sam_line(r) for r in recs if ("RG", "Z", "grpA") in r.tags)
    # golden BAM fixtures decode to their SAM twins
    for ds in MANIFEST["datasets"].values():
        out = subprocess.run([str(exe), str(GOLD / ds["bam"])], capture_output=True, text=True, check=True).stdout
        want = "".join(ln for ln in (GOLD / ds["sam"]).read_text().splitlines(True) if not ln.startswith("@"))
        assert out == want
    # corrupt input is diagnosed
    bad = tmp_path / "bad.bam"
    data = bytearray((tmp_path / "x1.bam").read_bytes())
    data[len(data) // 2] ^= 0xFF
    bad.write_bytes(bytes(data))
    pr = subprocess.run([str(exe), str(bad)], capture_output=True, text=True)
    assert pr.returncode != 0 and "bam2sam:" in pr.stderr


def test_bam_reader_edge_inputs(host, tmp_path):
    """header-only BAM, truncated files, a record larger than the batch buffer, input through a pipe"""
    _, pkg = host
    exe = pkg.PKG_DIR / "bin" / "bam2sam"
    _, refs, recs = tl.fuzz_dataset(77, 400)
    want = "".join(tl.sam_line(r) for r in recs).replace("\tchrNotInHeader\t", "\t*\t")

    empty = tmp_path / "hdr_only.bam"
    tl.write_bam(empty, refs, [])
    pr = subprocess.run([str(exe), str(empty)], capture_output=True, text=True)
    assert pr.returncode == 0 and pr.stdout == ""

    zero = tmp_path / "zero.bam"
    zero.write_bytes(b"")
    pr = subprocess.run([str(exe), str(zero)], capture_output=True, text=True)
    assert pr.returncode != 0 and "too short" in pr.stderr

    good = tmp_path / "good.bam"
    tl.write_bam(good, refs, recs, level=1, rng=np.random.default_rng(5), block=3000)
    data = good.read_bytes()
    # cut inside a BGZF block / cut on a block boundary but inside a record
    cut = tmp_path / "cut.bam"
    cut.write_bytes(data[: len(data) // 2])
    pr = subprocess.run([str(exe), str(cut)], capture_output=True, text=True)
    assert pr.returncode != 0 and "truncated" in pr.stderr
    raw = tl.bam_bytes(refs, recs)
    part = tmp_path / "part.bam"
    part.write_bytes(tl.bgzf_block(raw[: len(raw) - 7], 1) if len(raw) - 7 < 0xFF00 else
                     b"".join(tl.bgzf_block(raw[i:min(i + 0xFF00, len(raw) - 7)], 1) for i in range(0, len(raw) - 7, 0xFF00)))
    pr = subprocess.run([str(exe), str(part)], capture_output=True, text=True)
    assert pr.returncode != 0 and "truncated alignment record" in pr.stderr

    # not a regular file: the reader falls back to reading the stream into memory
    fifo = tmp_path / "in.fifo"
    os.mkfifo(fifo)
    feeder = subprocess.Popen(["sh", "-c", f"cat '{good}' > '{fifo}'"])
    pr = subprocess.run([str(exe), str(fifo)], capture_output=True, text=True, timeout=60)
    feeder.wait(timeout=60)
    assert pr.returncode == 0 and pr.stdout == want

    # one record longer than the (minimum, 256 KiB) batch buffer
    big = tl.Rec(qname="big", flag=0, rname=refs[0][0], pos=1, mapq=30, cigar=[(300000, "M")], seq="A" * 300000)
    bigf = tmp_path / "big.bam"
    tl.write_bam(bigf, refs, [recs[0], big, recs[1]])
    pr = subprocess.run([str(exe), str(bigf)], capture_output=True, text=True,
                        env={**os.environ, "PSSBAM_BATCH_BYTES": "262144"})
    assert pr.returncode != 0 and "exceeds the batch buffer" in pr.stderr
    pr = subprocess.run([str(exe), str(bigf)], capture_output=True, text=True)
    assert pr.returncode == 0 and pr.stdout.count("\n") == 3


def test_inflate_fast_and_crc_against_zlib(host):
    """pss_inflate_raw / pss_crc32 (the BGZF decoder of the feed) against zlib: stored, fixed and
    dynamic blocks from every level/strategy, skewed alphabets (codes longer than the table index),
    exact-size contract, truncated and bit-flipped streams (no stray writes, never more lenient than
    zlib on a stream zlib accepts)."""
    import zlib
    L, _ = host
    L.pss_inflate_raw.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.pss_inflate_raw.restype = C.c_int
    L.pss_crc32.argtypes = [C.c_uint32, C.c_char_p, C.c_size_t]
    L.pss_crc32.restype = C.c_uint32
    st = C.create_string_buffer(32768)
    rng = np.random.default_rng(1)

    def raw_deflate(data, level, strategy, memlevel):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
        return c.compress(data) + c.flush()

    def gen(kind, n):
        if kind == 0:
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        if kind == 1:
            return rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes()
        if kind == 2:
            return bytes(n)
        if kind == 3:   # skewed alphabet with rare symbols -> codewords longer than 11 bits
            p = np.array([2.0 ** -(i * 0.35) for i in range(200)])
            return rng.choice(np.arange(200, dtype=np.uint8), n, p=p / p.sum()).tobytes()
        if kind == 4:   # repeats at all distances, runs (distance 1), overlapping copies
            base = rng.integers(0, 256, 300, dtype=np.uint8).tobytes()
            out = bytearray()
            while len(out) < n:
                k = int(rng.integers(0, 280))
                out += base[k:k + int(rng.integers(3, 300))]
                if rng.random() < 0.3:
                    out += bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 400))
            return bytes(out[:n])
        return b"".join(b"read%07d\t%d\tchr%d\t%d\t37\t100M\t*\t0\t0\n" % (i, i % 5, i % 22, i * 13)
                        for i in range(n // 40 + 1))[:n]

    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]
    for it in range(400):
        n = int(rng.integers(0, 65536 if it % 3 else 2000))
        data = gen(int(rng.integers(0, 6)), n)
        n = len(data)
        comp = raw_deflate(data, int(rng.integers(0, 10)), strategies[int(rng.integers(0, 5))], int(rng.integers(1, 10)))
        out = C.create_string_buffer(n + 16)
        out.raw = b"\xAA" * (n + 16)
        assert L.pss_inflate_raw(st, comp, len(comp), out, n) == 0, it
        assert out.raw[:n] == data and out.raw[n:] == b"\xAA" * 16, it
        assert L.pss_crc32(0, data, n) == zlib.crc32(data)
        k = int(rng.integers(0, n + 1))
        assert L.pss_crc32(L.pss_crc32(0, data[:k], k), data[k:], n - k) == zlib.crc32(data)
        if n:
            assert L.pss_inflate_raw(st, comp, len(comp), out, n - 1) != 0       # stream longer than promised
        assert L.pss_inflate_raw(st, comp, len(comp), out, n + 1) != 0           # stream shorter than promised
        cut = int(rng.integers(0, len(comp)))
        assert L.pss_inflate_raw(st, comp[:cut], cut, out, n) != 0               # truncated input
        if len(comp) > 4:
            b = bytearray(comp)
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            rc = L.pss_inflate_raw(st, bytes(b), len(b), out, n)
            try:
                d = zlib.decompressobj(-15)
                z = d.decompress(bytes(b))
                zok = d.eof and len(z) == n
            except zlib.error:
                zok = False
            if zok:
                assert rc == 0 and out.raw[:n] == z
            else:
                assert rc != 0, "accepted a stream zlib rejects"
        assert out.raw[n:] == b"\xAA" * 16, "wrote past the end on an error path"


def test_front_end_argument_handling(host, tmp_path):
    _, pkg = host
    pss, fk = pkg.PKG_DIR / "bin" / "pss-bam", pkg.PKG_DIR / "bin" / "fragkon"
    pr = subprocess.run([str(pss)], capture_output=True, text=True)
    assert pr.returncode == 1 and pr.stderr.startswith("pss-bam v1.2.1: Program for describing base context")
    pr = subprocess.run([str(pss), "-F"], capture_output=True, text=True)
    assert pr.returncode == 0 and "Please enter required argument for option -F." in pr.stderr
    pr = subprocess.run([str(fk), "-B", "x"], capture_output=True, text=True)
    assert pr.returncode == 1 and pr.stderr.startswith("fragkon: Program for describing kmer-based")
    pr = subprocess.run([str(pss), "-F", str(tmp_path / "nope.fa"), "-B", "x.bam", "-o", "o"], capture_output=True, text=True)
    assert pr.returncode == 1 and "Reading genome sequence from:" in pr.stderr and "Cannot open file" in pr.stderr


def test_sam_reader_encodes_what_line2saml_accepts(host, tmp_path):
    """SAM text -> BAM records -> text again: flag/rname/pos/mapq/cigar/seq/qual survive, RG is
    kept, lines line2saml rejects are dropped and counted, non-canonical CIGAR text becomes '*'"""
    L, pkg = host
    _, refs, recs = tl.fuzz_dataset(77, 800, with_rg=True)
    sam = tmp_path / "in.sam"
    tl.write_sam(sam, refs, recs)
    with open(sam, "a") as fh:
        fh.write("short\t0\tchrA\t5\n")                                              # < 11 fields
        fh.write("lenmis\t0\tchrA\t5\t30\t3M\t*\t0\t0\tACG\tII\n")                 # SEQ/QUAL mismatch
        fh.write("noncanon\t0\tchrA\t5\t30\t03M\t*\t0\t0\tACG\tIII\n")             # "03M" never equals "%dM"
        fh.write("lower\t16\tchrNew\t7\t300\t3M\t*\t0\t0\tacn\tIII\tXX:i:1\tRG:Z:g9\n")
    L.sam_reader_open.restype = C.c_void_p
    L.sam_reader_open.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.sam_reader_next.restype = C.c_int64
    L.sam_reader_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.sam_reader_n_ref.argtypes = [C.c_void_p]
    L.sam_reader_ref_names.restype = C.POINTER(C.c_char_p)
    L.sam_reader_ref_names.argtypes = [C.c_void_p]
    L.sam_reader_lines_skipped.restype = C.c_uint64
    L.sam_reader_lines_skipped.argtypes = [C.c_void_p]
    L.sam_reader_close.argtypes = [C.c_void_p]

    class Hdr(C.Structure):
        _fields_ = [("text", C.c_char_p), ("l_text", C.c_uint32), ("n_ref", C.c_int32), ("ref_name", C.POINTER(C.c_char_p)),
                    ("ref_len", C.c_void_p)]
    L.bam_record_to_sam.restype = C.c_long
    L.bam_record_to_sam.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(Hdr), C.c_char_p, C.c_size_t]
    err = C.create_string_buffer(256)
    rd = L.sam_reader_open(str(sam).encode(), 1 << 20, err, 256)
    assert rd, err.value
    lines = []
    out = C.create_string_buffer(1 << 16)
    while True:
        recs_p, offs_p, nb = C.c_void_p(), C.c_void_p(), C.c_size_t()
        n = L.sam_reader_next(rd, C.byref(recs_p), C.byref(offs_p), C.byref(nb))
        assert n >= 0
        if n == 0:
            break
        offs = np.ctypeslib.as_array(C.cast(offs_p, C.POINTER(C.c_uint32)), shape=(n + 1,)).copy()
        hdr = Hdr(None, 0, L.sam_reader_n_ref(rd), L.sam_reader_ref_names(rd), None)
        for i in range(n):
            w = L.bam_record_to_sam(recs_p.value + int(offs[i]), int(offs[i + 1] - offs[i]), C.byref(hdr), out, 1 << 16)
            assert w > 0
            lines.append(out.value.decode())
    assert L.sam_reader_lines_skipped(rd) == 2 + sum(1 for r in recs if len(r.seq) != len(r.qual))
    L.sam_reader_close(rd)
    kept = [r for r in recs if len(r.seq) == len(r.qual)]
    assert len(lines) == len(kept) + 2
    for r, ln in zip(kept, lines):
        f = ln.rstrip("\n").split("\t")
        assert (f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5]) == (r.qname, r.flag, r.rname, r.pos, r.mapq, r.cigar_str())
        want_seq = "*" if r.seq == "*" else "".join(c if c in "=ACMGRSVTWYHKDBN" else "N" for c in r.seq.upper())
        assert f[9] == want_seq and f[10] == r.qual
        assert int(f[8]) == (r.tlen if r.flag & 1 else 0)
        rg = [t for t in r.tags if t[0] == "RG"]
        assert f[11:] == ([f"RG:Z:{rg[0][2]}"] if rg else [])
    f = lines[-2].split("\t")
    assert f[0] == "noncanon" and f[5] == "*"
    f = lines[-1].rstrip("\n").split("\t")
    assert f[0] == "lower" and f[2] == "chrNew" and f[4] == "255" and f[9] == "ACN" and f[11:] == ["RG:Z:g9"]


def _sam_reader_all(L, path, batch):
    """all encoded record bytes + the reference names the reader ends with + skipped count"""
    L.sam_reader_open.restype = C.c_void_p
    L.sam_reader_open.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.sam_reader_next.restype = C.c_int64
    L.sam_reader_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.sam_reader_n_ref.argtypes = [C.c_void_p]
    L.sam_reader_ref_names.restype = C.POINTER(C.c_char_p)
    L.sam_reader_ref_names.argtypes = [C.c_void_p]
    L.sam_reader_lines_skipped.restype = C.c_uint64
    L.sam_reader_lines_skipped.argtypes = [C.c_void_p]
    L.sam_reader_close.argtypes = [C.c_void_p]
    err = C.create_string_buffer(256)
    rd = L.sam_reader_open(str(path).encode(), batch, err, 256)
    assert rd, err.value
    chunks, n_rec, n_batches = [], 0, 0
    while True:
        recs_p, offs_p, nb = C.c_void_p(), C.c_void_p(), C.c_size_t()
        n = L.sam_reader_next(rd, C.byref(recs_p), C.byref(offs_p), C.byref(nb))
        assert n >= 0
        if n == 0:
            break
        offs = np.ctypeslib.as_array(C.cast(offs_p, C.POINTER(C.c_uint32)), shape=(n + 1,))
        assert offs[0] == 0 and offs[n] == nb.value and np.all(np.diff(offs.astype(np.int64)) >= 36)
        chunks.append(C.string_at(recs_p.value, nb.value))
        n_rec += n
        n_batches += 1
    names = [L.sam_reader_ref_names(rd)[i].decode() for i in range(L.sam_reader_n_ref(rd))]
    skipped = L.sam_reader_lines_skipped(rd)
    L.sam_reader_close(rd)
    return b"".join(chunks), n_rec, names, skipped, n_batches


def test_sam_reader_threads_and_batch_seams(host, tmp_path):
    """the multi-threaded text parser gives byte-identical records, names and counts whatever the
    thread count and batch size (pieces and batches are cut at line starts), from plain and gzip
    text, with RNAMEs the header never announced and a line longer than fgets' MAX_LINE_LEN"""
    import gzip
    L, _ = host
    _, refs, recs = tl.fuzz_dataset(91, 30000, with_rg=True)
    sam = tmp_path / "big.sam"
    tl.write_sam(sam, refs, recs)
    with open(sam, "a") as fh:
        fh.write("late1\t0\tchrLateA\t5\t30\t3M\t*\t0\t0\tACG\tIII\n")
        fh.write("x" * 450000 + "\t0\tchrA\t5\t30\t3M\t*\t0\t0\tACG\tIII\n")   # cut into 200000-char "lines": all rejected
        fh.write("late2\t16\tchrLateB\t9\t30\t2M\t*\t0\t0\tAC\tII\n")
        fh.write("late3\t0\tchrLateA\t6\t30\t3M\t*\t0\t0\tACG\tIII")              # no final newline
    gz = tmp_path / "big.sam.gz"
    gz.write_bytes(gzip.compress(sam.read_bytes(), 1))
    assert sam.stat().st_size > (4 << 20)
    want = None
    for threads, batch, path in (("1", 0, sam), ("7", 1 << 20, sam), ("16", 3 << 20, sam), ("5", 1 << 20, gz), ("1", 0, gz)):
        os.environ["PSSBAM_SAM_THREADS"] = threads
        try:
            got = _sam_reader_all(L, path, batch)
        finally:
            del os.environ["PSSBAM_SAM_THREADS"]
        if want is None:
            want = got
            assert got[1] == sum(1 for r in recs if len(r.seq) == len(r.qual)) + 3
            assert got[2][-2:] == ["chrLateA", "chrLateB"] and got[3] >= 3
        else:
            assert got[:4] == want[:4], (threads, batch, path.name)
        if batch:
            assert got[4] > 3   # really several batches


def test_bam_reader_block_layouts(host, tmp_path):
    """the same records in htslib's block layout (every BGZF block starts on a record boundary: the
    reader takes the inflate workers' per-block record lists) and cut every 0xff00 bytes regardless
    of records (serial chain walk), small and default batches: identical text, equal to the
    model's independently written SAM twin"""
    _, pkg = host
    from pss_bam_amd import synth
    exe = pkg.PKG_DIR / "bin" / "bam2sam"
    d = synth.config("C4", n_reads=150_000, scale_genome=0.0005)
    d.pop("region_len")
    cfg = synth.make_cfg(**d)
    n = int(cfg.n_reads)
    sam = tmp_path / "twin.sam"
    synth.sam_host(cfg, 0, n, sam, with_header=False)
    want = sam.read_text()
    for ragged in (False, True):
        bam = tmp_path / f"layout{int(ragged)}.bam"
        synth.bam_file_host(cfg, 0, n, bam, level=1, threads=4, ragged=ragged)
        for batch in ("0", "262144", "1048576"):
            env = dict(os.environ)
            if batch != "0":
                env["PSSBAM_BATCH_BYTES"] = batch
            got = subprocess.run([str(exe), str(bam)], capture_output=True, text=True, check=True, env=env).stdout
            assert got == want, (ragged, batch)


def test_kmer_count_saturates_at_uint_max_on_print(host, tmp_path):
    """The reference's k-mer bins are `unsigned int` that stick at UINT_MAX (kmer.c:102-104); the
    device bins are u64 and the printers clamp.  A bin of 2^32+5 must print as 4294967295 in both
    tables (fragkon.c:231-249, genome-kmer-count.c:56-66), its neighbours untouched."""
    L, _ = host
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.fragkon_write_table.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    L.gkc_write_table.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    k = 3
    k5 = np.arange(4 ** k, dtype=np.uint64)
    k3 = np.zeros(4 ** k, dtype=np.uint64)
    k5[7] = (1 << 32) + 5
    k5[8] = (1 << 32) - 1            # exactly UINT_MAX: printed as is
    k3[9] = 1 << 40
    k3[10] = (1 << 32) - 2
    f = libc.fopen(str(tmp_path / "fk.txt").encode(), b"w")
    assert L.fragkon_write_table(f, b"g.fa", b"a.bam", k, k5.ctypes.data, k3.ctypes.data) == 0
    libc.fclose(f)
    got5, got3 = tl.parse_fragkon_text((tmp_path / "fk.txt").read_text())
    want5, want3 = np.minimum(k5, 0xFFFFFFFF).astype(np.uint32), np.minimum(k3, 0xFFFFFFFF).astype(np.uint32)
    assert np.array_equal(got5, want5) and np.array_equal(got3, want3)
    lines = (tmp_path / "fk.txt").read_text().splitlines()
    assert lines[4 + 7] == "ACT\t4294967295\t0" and lines[4 + 9] == "AGC\t9\t4294967295"
    f = libc.fopen(str(tmp_path / "gkc.txt").encode(), b"w")
    assert L.gkc_write_table(f, k, k5.ctypes.data) == 0
    libc.fclose(f)
    rows = [ln.split("\t") for ln in (tmp_path / "gkc.txt").read_text().splitlines()]
    assert [int(c) for _, c in rows] == [int(x) for x in want5]
    assert rows[7] == ["ACT", "4294967295"] and rows[6] == ["ACG", "6"]
