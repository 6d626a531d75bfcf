"""Shared test/bench helpers (test infrastructure, not product code).

* builds + wraps ``oracle/liboracle.so`` (the CPU restatement) through ctypes,
* runs the real reference binaries in ``oracle/_ref/`` behind the PATH shim,
* writes FASTA / SAM / BAM files from one Python record list.  The SAM writer and the
  BAM writer are *independent* encoders of the same ``Rec`` objects (avoid a
  common-mode decoder bug between the oracle's text input and the engine's binary one),
* seeded random genomes / alignments for fuzzing.

Nothing here reads /root/reference at run time; ``oracle/_ref`` is a prebuilt artefact.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess
import zlib
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
REF_DIR = ORACLE_DIR / "_ref"
SHIM_DIR = ORACLE_DIR / "shim"

# ----------------------------------------------------------------------------------
# building
# ----------------------------------------------------------------------------------


def build_oracle() -> Path:
    """make -C oracle (restatement always; _ref only when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", str(ORACLE_DIR)], check=True, stdout=subprocess.DEVNULL)
    return ORACLE_DIR / "liboracle.so"


def have_ref() -> bool:
    return (REF_DIR / "pss-bam").exists() and (REF_DIR / "fragkon").exists()


# ----------------------------------------------------------------------------------
# oracle (ctypes)
# ----------------------------------------------------------------------------------

ST_OK, ST_PARSE_SKIP, ST_NO_CONTIG, ST_FILTERED, ST_KMER_FAIL, ST_N = 0, 1, 2, 3, 4, 5


class _PssParams(C.Structure):
    _fields_ = [("region_len", C.c_int), ("min_read_len", C.c_ulong), ("max_read_len", C.c_ulong),
                ("min_mq", C.c_int), ("up_ctx", C.c_char_p), ("down_ctx", C.c_char_p),
                ("merged_only", C.c_int)]


class _FkParams(C.Structure):
    _fields_ = [("klen", C.c_int), ("min_mq", C.c_int), ("min_read_len", C.c_ulong),
                ("max_read_len", C.c_ulong), ("merged_only", C.c_int)]


@dataclass
class PssOpts:
    """pss-bam command-line options (pss-bam.c:12-18 defaults)."""
    region_len: int = 15
    min_read_len: int = 0
    max_read_len: int = 250000000
    min_mq: int = 0
    up_ctx: str = "ACGT"
    down_ctx: str = "ACGT"
    merged_only: bool = False
    read_group: str | None = None

    def argv(self) -> list[str]:
        a = ["-r", str(self.region_len), "-l", str(self.min_read_len), "-L", str(self.max_read_len),
             "-q", str(self.min_mq), "-U", self.up_ctx, "-D", self.down_ctx]
        if self.merged_only:
            a.append("-m")
        if self.read_group is not None:
            a += ["-R", self.read_group]
        return a


@dataclass
class FkOpts:
    """fragkon command-line options (fragkon.c:14-18 defaults)."""
    klen: int = 8
    min_mq: int = 0
    min_read_len: int = 0
    max_read_len: int = 250000000
    merged_only: bool = False

    def argv(self) -> list[str]:
        a = ["-k", str(self.klen), "-l", str(self.min_read_len), "-L", str(self.max_read_len),
             "-q", str(self.min_mq)]
        if self.merged_only:
            a.append("-m")
        return a


class Oracle:
    """ctypes face of oracle/liboracle.so."""

    def __init__(self):
        self.lib = C.CDLL(str(build_oracle()))
        L = self.lib
        L.orc_genome_load.restype = C.c_void_p
        L.orc_genome_load.argtypes = [C.c_char_p]
        L.orc_genome_from_arrays.restype = C.c_void_p
        L.orc_genome_from_arrays.argtypes = [C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_size_t)]
        L.orc_genome_free.argtypes = [C.c_void_p]
        L.orc_pss_run.restype = C.c_int
        L.orc_pss_run.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_PssParams), C.c_void_p, C.c_void_p,
                                  C.c_void_p]
        L.orc_fk_run.restype = C.c_int
        L.orc_fk_run.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_FkParams), C.c_void_p, C.c_void_p,
                                 C.c_void_p]
        L.orc_genome_kmer_count.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_pss_rates.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pss_write_counts.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pss_write_rates.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]

    def load_genome(self, fasta: str | Path):
        g = self.lib.orc_genome_load(str(fasta).encode())
        if not g:
            raise RuntimeError(f"oracle could not load {fasta}")
        return g

    def genome_from_arrays(self, contigs: list[tuple[str, np.ndarray]]):
        n = len(contigs)
        ids = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
        seqs = (C.c_void_p * n)(*[c[1].ctypes.data for c in contigs])
        lens = (C.c_size_t * n)(*[c[1].size for c in contigs])
        return self.lib.orc_genome_from_arrays(n, ids, seqs, lens)

    def free_genome(self, g):
        self.lib.orc_genome_free(g)

    def pss(self, genome, sam: str | Path, o: PssOpts):
        """-> (fwd[(N+2),16] u64, rev, status[ST_N])"""
        p = _PssParams(o.region_len, o.min_read_len, o.max_read_len, o.min_mq, o.up_ctx.encode(),
                       o.down_ctx.encode(), int(o.merged_only))
        n = o.region_len + 2
        fwd = np.zeros((n, 16), dtype=np.uint64)
        rev = np.zeros((n, 16), dtype=np.uint64)
        st = np.zeros(ST_N, dtype=np.uint64)
        rc = self.lib.orc_pss_run(genome, str(sam).encode(), C.byref(p), fwd.ctypes.data, rev.ctypes.data,
                                  st.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_pss_run failed rc={rc}")
        return fwd, rev, st

    def fragkon(self, genome, sam: str | Path, o: FkOpts):
        """-> (k5[4^k] u32, k3, status)"""
        p = _FkParams(o.klen, o.min_mq, o.min_read_len, o.max_read_len, int(o.merged_only))
        nb = 4 ** o.klen
        k5 = np.zeros(nb, dtype=np.uint32)
        k3 = np.zeros(nb, dtype=np.uint32)
        st = np.zeros(ST_N, dtype=np.uint64)
        rc = self.lib.orc_fk_run(genome, str(sam).encode(), C.byref(p), k5.ctypes.data, k3.ctypes.data,
                                 st.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_fk_run failed rc={rc}")
        return k5, k3, st

    def genome_kmer_count(self, genome, klen: int) -> np.ndarray:
        out = np.zeros(4 ** klen, dtype=np.uint32)
        rc = self.lib.orc_genome_kmer_count(genome, klen, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"orc_genome_kmer_count failed rc={rc}")
        return out

    def rates(self, counts: np.ndarray) -> np.ndarray:
        n = counts.shape[0] - 2
        r = np.zeros((n, 12), dtype=np.float64)
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        self.lib.orc_pss_rates(n, c.ctypes.data, r.ctypes.data)
        return r

    def write_reports(self, fasta_fn: str, bam_fn: str, prefix: str, fwd: np.ndarray, rev: np.ndarray):
        n = fwd.shape[0] - 2
        f = np.ascontiguousarray(fwd, dtype=np.uint64)
        r = np.ascontiguousarray(rev, dtype=np.uint64)
        self.lib.orc_pss_write_counts(fasta_fn.encode(), bam_fn.encode(), prefix.encode(), n, f.ctypes.data,
                                      r.ctypes.data)
        fr, rr = self.rates(f), self.rates(r)
        self.lib.orc_pss_write_rates(fasta_fn.encode(), bam_fn.encode(), prefix.encode(), n, fr.ctypes.data,
                                     rr.ctypes.data)


# ----------------------------------------------------------------------------------
# running the real reference (oracle/_ref) behind the samtools PATH shim
# ----------------------------------------------------------------------------------


def _ref_env(bam2sam: str | None = None) -> dict:
    env = dict(os.environ)
    env["PATH"] = f"{SHIM_DIR}:{env.get('PATH', '')}"
    if bam2sam:
        env["PSSBAM_BAM2SAM"] = str(bam2sam)
    return env


def parse_counts_text(text: str) -> tuple[np.ndarray, np.ndarray]:
    """.pss.counts.txt -> (fwd, rev) in the in-memory row order of the reference
    (row0 = 2nd context base, row1 = 1st, row 2+i = position i)."""
    blocks = text.split("### Reverse read substitution counts and base context\n")
    assert len(blocks) == 2, "unexpected counts file layout"

    def rows(b):
        out = []
        for ln in b.splitlines():
            if not ln or ln.startswith("#"):
                continue
            parts = ln.split("\t")
            assert parts[-1] == "", "row must end with a trailing TAB"
            out.append((int(parts[0]), [int(x) for x in parts[1:17]]))
        return out

    f = rows(blocks[0])
    r = rows(blocks[1])
    n = len(f) - 2
    fwd = np.zeros((n + 2, 16), dtype=np.uint64)
    rev = np.zeros((n + 2, 16), dtype=np.uint64)
    for k, (label, vals) in enumerate(f):
        assert label == k - 2
        fwd[k] = vals
    # reverse block: N-1..0, then rows labelled 1 (=row 1) and 2 (=row 0)
    for k, (label, vals) in enumerate(r[:n]):
        assert label == n - 1 - k
        rev[label + 2] = vals
    assert r[n][0] == 1 and r[n + 1][0] == 2
    rev[1] = r[n][1]
    rev[0] = r[n + 1][1]
    return fwd, rev


def run_ref_pss(fasta: Path, aln: Path, prefix: Path, o: PssOpts, variant: str = "pss-bam",
                bam2sam: str | None = None, timeout: float = 600.0):
    """Runs oracle/_ref/<variant>; returns (fwd, rev, counts_text, rates_text, stderr)."""
    exe = REF_DIR / variant
    cmd = [str(exe), "-F", str(fasta), "-B", str(aln), "-o", str(prefix)] + o.argv()
    pr = subprocess.run(cmd, env=_ref_env(bam2sam), capture_output=True, text=True, timeout=timeout)
    if pr.returncode != 0:
        raise RuntimeError(f"reference pss-bam failed ({pr.returncode}): {pr.stderr[-2000:]}")
    ct = Path(f"{prefix}.pss.counts.txt").read_text()
    rt = Path(f"{prefix}.pss.rates.txt").read_text()
    fwd, rev = parse_counts_text(ct)
    return fwd, rev, ct, rt, pr.stderr


def run_ref_gkc(fasta: Path, klen: int, timeout: float = 600.0) -> tuple[np.ndarray, str]:
    """oracle/_ref/genome-kmer-count -> (counts[4^k] u32, stdout)"""
    pr = subprocess.run([str(REF_DIR / "genome-kmer-count"), "-f", str(fasta), "-k", str(klen)], capture_output=True,
                        text=True, timeout=timeout)
    if pr.returncode != 0:
        raise RuntimeError(f"reference genome-kmer-count failed ({pr.returncode}): {pr.stderr[-1000:]}")
    rows = [ln.split("\t") for ln in pr.stdout.splitlines()[1:]]
    return np.array([int(c) for _, c in rows], dtype=np.uint32), pr.stdout


def parse_fragkon_text(text: str) -> tuple[np.ndarray, np.ndarray]:
    k5, k3 = [], []
    for ln in text.splitlines():
        if ln.startswith("#"):
            continue
        _, a, b = ln.split("\t")
        k5.append(int(a))
        k3.append(int(b))
    return np.array(k5, dtype=np.uint32), np.array(k3, dtype=np.uint32)


def run_ref_fragkon(fasta: Path, aln: Path, o: FkOpts, variant: str = "fragkon", bam2sam: str | None = None,
                    timeout: float = 600.0):
    exe = REF_DIR / variant
    # fragkon.c:372 free()s a pointer it never initialised; depending on stack garbage glibc
    # aborts the process *after* the table was printed but before stdio is flushed.  Line-buffer
    # stdout (stdbuf) so the table is complete either way, and accept that abort iff it is.
    cmd = ["stdbuf", "-oL", str(exe), "-F", str(fasta), "-B", str(aln)] + o.argv()
    pr = subprocess.run(cmd, env=_ref_env(bam2sam), capture_output=True, text=True, timeout=timeout)
    n_rows = sum(1 for ln in pr.stdout.splitlines() if not ln.startswith("#"))
    complete = n_rows == 4 ** o.klen and pr.stdout.endswith("\n")
    if pr.returncode != 0 and not (complete and pr.returncode in (-6, -11)):
        raise RuntimeError(f"reference fragkon failed ({pr.returncode}): {pr.stderr[-2000:]}")
    k5, k3 = parse_fragkon_text(pr.stdout)
    return k5, k3, pr.stdout, pr.stderr


# ----------------------------------------------------------------------------------
# alignment records and the two independent writers
# ----------------------------------------------------------------------------------

CIGAR_OPS = "MIDNSHP=X"
SEQ_CODES = "=ACMGRSVTWYHKDBN"


@dataclass
class Rec:
    qname: str
    flag: int
    rname: str          # contig name or '*'
    pos: int            # 1-based POS (0 = unavailable)
    mapq: int
    cigar: list         # [(len, 'M'), ...]; [] means '*'
    tlen: int = 0
    seq: str = "*"      # '*' = absent
    qual: str = "*"     # '*' = absent (BAM 0xFF fill)
    rnext: str = "*"
    pnext: int = 0
    tags: list = field(default_factory=list)   # [("RG", "Z", "grp1"), ("NM", "i", 3)]

    def cigar_str(self) -> str:
        return "".join(f"{n}{op}" for n, op in self.cigar) if self.cigar else "*"

    def ref_span(self) -> int:
        return sum(n for n, op in self.cigar if op in "MDN=X")


def sam_line(r: Rec) -> str:
    f = [r.qname, str(r.flag), r.rname, str(r.pos), str(r.mapq), r.cigar_str(), r.rnext, str(r.pnext),
         str(r.tlen), r.seq, r.qual]
    for tag, typ, val in r.tags:
        if typ == "B":                      # (sub-type, values): TAG:B:<sub>,v,v,...
            sub, vals = val
            f.append(f"{tag}:B:{sub}" + "".join(f",{v:g}" if sub == "f" else f",{int(v)}" for v in vals))
        elif typ == "f":
            f.append(f"{tag}:f:{float(val):g}")
        else:
            f.append(f"{tag}:{typ}:{val}")
    return "\t".join(f) + "\n"


def write_sam(path: Path, refs: list[tuple[str, int]], recs: list[Rec], header: bool = True) -> None:
    with open(path, "w") as fh:
        if header:
            fh.write("@HD\tVN:1.6\tSO:unknown\n")
            for name, ln in refs:
                fh.write(f"@SQ\tSN:{name}\tLN:{ln}\n")
        for r in recs:
            fh.write(sam_line(r))


def _reg2bin(beg: int, end: int) -> int:
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def bam_record(r: Rec, ref_index: dict[str, int]) -> bytes:
    """One BAM alignment record (with its leading block_size), SAM spec section 4.2."""
    ref_id = ref_index.get(r.rname, -1) if r.rname != "*" else -1
    nref_id = ref_id if r.rnext == "=" else (ref_index.get(r.rnext, -1) if r.rnext != "*" else -1)
    name = r.qname.encode() + b"\0"
    cig = b"".join(struct.pack("<I", (n << 4) | CIGAR_OPS.index(op)) for n, op in r.cigar)
    if r.seq == "*":
        l_seq, seq_b, qual_b = 0, b"", b""
    else:
        l_seq = len(r.seq)
        codes = [SEQ_CODES.index(ch) if ch in SEQ_CODES else 15 for ch in r.seq.upper()]
        if l_seq & 1:
            codes.append(0)
        seq_b = bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
        qual_b = b"\xff" * l_seq if r.qual == "*" else bytes(ord(c) - 33 for c in r.qual)
        assert len(qual_b) == l_seq
    aux = b""
    for tag, typ, val in r.tags:
        t = tag.encode()
        if typ == "Z":
            aux += t + b"Z" + str(val).encode() + b"\0"
        elif typ == "A":
            aux += t + b"A" + str(val).encode()[:1]
        elif typ == "i":
            v = int(val)
            if 0 <= v < 256:
                aux += t + b"C" + struct.pack("<B", v)
            elif -128 <= v < 0:
                aux += t + b"c" + struct.pack("<b", v)
            elif 0 <= v < 65536:
                aux += t + b"S" + struct.pack("<H", v)
            elif -32768 <= v < 0:
                aux += t + b"s" + struct.pack("<h", v)
            elif v >= 1 << 31:
                aux += t + b"I" + struct.pack("<I", v)
            else:
                aux += t + b"i" + struct.pack("<i", v)
        elif typ == "f":
            aux += t + b"f" + struct.pack("<f", float(val))
        elif typ == "H":
            aux += t + b"H" + str(val).encode() + b"\0"
        elif typ == "B":
            sub, vals = val
            fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[sub]
            aux += t + b"B" + sub.encode() + struct.pack("<I", len(vals)) + struct.pack(f"<{len(vals)}{fmt}", *vals)
        else:
            raise ValueError(typ)
    pos0 = r.pos - 1
    end0 = pos0 + (r.ref_span() or 1)
    core = struct.pack("<iiBBHHHIiii", ref_id, pos0, len(name), r.mapq & 0xFF, _reg2bin(max(pos0, 0), max(end0, 1)),
                       len(r.cigar), r.flag & 0xFFFF, l_seq, nref_id, r.pnext - 1, r.tlen)
    body = core + name + cig + seq_b + qual_b + aux
    return struct.pack("<I", len(body)) + body


def bgzf_block(data: bytes, level: int = 6) -> bytes:
    assert len(data) <= 0xFF00
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    cdata = co.compress(data) + co.flush()
    bsize = len(cdata) + 25
    hdr = struct.pack("<BBBBIBBHBBHH", 0x1F, 0x8B, 8, 4, 0, 0, 0xFF, 6, ord("B"), ord("C"), 2, bsize)
    return hdr + cdata + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def bam_bytes(refs: list[tuple[str, int]], recs: list[Rec], text_header: str | None = None) -> bytes:
    """Uncompressed BAM stream (magic + header + references + records)."""
    if text_header is None:
        text_header = "@HD\tVN:1.6\tSO:unknown\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    th = text_header.encode()
    out = [b"BAM\1", struct.pack("<i", len(th)), th, struct.pack("<i", len(refs))]
    for n, l in refs:
        nb = n.encode() + b"\0"
        out += [struct.pack("<i", len(nb)), nb, struct.pack("<i", l)]
    idx = {n: i for i, (n, _) in enumerate(refs)}
    out += [bam_record(r, idx) for r in recs]
    return b"".join(out)


def write_bam(path: Path, refs: list[tuple[str, int]], recs: list[Rec], level: int = 6, block: int = 0xFF00,
              rng: np.random.Generator | None = None) -> None:
    """BGZF-compressed BAM.  With `rng`, block payload sizes are randomised so records
    (and even their block_size words) straddle BGZF block boundaries."""
    raw = bam_bytes(refs, recs)
    with open(path, "wb") as fh:
        i = 0
        while i < len(raw):
            n = block if rng is None else int(rng.integers(1, block + 1))
            fh.write(bgzf_block(raw[i:i + n], level))
            i += n
        fh.write(BGZF_EOF)


def write_bam_aligned(path: Path, refs: list[tuple[str, int]], recs: list[Rec], level: int = 6, block: int = 0xFF00,
                      rng: np.random.Generator | None = None) -> int:
    """BGZF BAM in htslib's layout: the header in its own block(s), every later block holding whole
    records only (a block is flushed before a record that would not fit; with `rng` at random fill
    levels).  Returns the number of header bytes in the inflated stream."""
    head = bam_bytes(refs, [])
    idx = {n: i for i, (n, _) in enumerate(refs)}
    with open(path, "wb") as fh:
        for i in range(0, len(head), block):
            fh.write(bgzf_block(head[i:i + block], level))
        cur, limit = [], block if rng is None else int(rng.integers(300, block + 1))
        size = 0
        for r in recs:
            b = bam_record(r, idx)
            assert len(b) <= block
            if size + len(b) > limit and cur:
                fh.write(bgzf_block(b"".join(cur), level))
                cur, size = [], 0
                limit = block if rng is None else int(rng.integers(300, block + 1))
            cur.append(b)
            size += len(b)
        if cur:
            fh.write(bgzf_block(b"".join(cur), level))
        fh.write(BGZF_EOF)
    return len(head)


# ----------------------------------------------------------------------------------
# FASTA
# ----------------------------------------------------------------------------------


def write_fasta(path: Path, contigs: list[tuple[str, str]], width: int = 60, descr: bool = True,
                gz: bool = False) -> None:
    """contigs: [(id, sequence-as-written)], sequence may hold lower case / IUPAC."""
    chunks = []
    for k, (cid, seq) in enumerate(contigs):
        chunks.append(f">{cid} synthetic contig {k}\n" if descr else f">{cid}\n")
        for i in range(0, len(seq), width):
            chunks.append(seq[i:i + width] + "\n")
    data = "".join(chunks).encode()
    if gz:
        import gzip
        with gzip.open(path, "wb") as fh:
            fh.write(data)
    else:
        Path(path).write_bytes(data)


def random_contig(rng: np.random.Generator, n: int, lower_frac: float = 0.3, n_frac: float = 0.01,
                  iupac_frac: float = 0.002) -> str:
    a = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n)
    # runs of N
    for _ in range(max(1, int(n * n_frac / 8))):
        p = int(rng.integers(0, n))
        a[p:p + int(rng.integers(1, 16))] = ord("N")
    m = rng.random(n) < iupac_frac
    a[m] = rng.choice(np.frombuffer(b"RYKMSWN", dtype=np.uint8), size=int(m.sum()))
    # soft-masked stretches
    i = 0
    while i < n:
        run = int(rng.integers(20, 200))
        if rng.random() < lower_frac:
            a[i:i + run] |= 0x20
        i += run
    return a.tobytes().decode()


_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


_B_RANGE = {"c": (-128, 128), "C": (0, 256), "s": (-32768, 32768), "S": (0, 65536), "i": (-(1 << 31), 1 << 31),
            "I": (0, 1 << 32)}


def random_aux(rng: np.random.Generator, max_count: int = 24) -> tuple:
    """one optional field of a type the -R walk has to step over (SAM spec 4.2.4): B arrays of every
    sub-type (count 0 included), H, f, the small / negative / large integer encodings, and values whose
    bytes spell an RG:Z field themselves"""
    u = int(rng.integers(0, 10))
    if u < 3:
        sub = "cCsSiIf"[int(rng.integers(0, 7))]
        n = 0 if rng.random() < 0.2 else int(rng.integers(1, max_count + 1))
        if sub == "f":
            vals = [float(np.float32(x)) for x in rng.normal(0, 50, size=n)]
        else:
            lo, hi = _B_RANGE[sub]
            vals = [int(x) for x in rng.integers(lo, hi, size=n)]
        return ("ML" if sub == "C" else "Z" + sub, "B", (sub, vals))
    if u == 3:
        return ("XH", "H", "".join("0123456789ABCDEF"[int(x)] for x in rng.integers(0, 16, size=2 * int(rng.integers(0, 12)))))
    if u == 4:
        return ("XF", "f", float(np.float32(rng.normal(0, 1000))))
    if u == 5:
        return ("XS", "i", int(rng.integers(-32768, -128)))            # 's' in BAM
    if u == 6:
        return ("XI", "i", int(rng.integers(1 << 31, 1 << 32)))         # 'I' in BAM
    if u == 7:
        return ("XW", "i", int(rng.integers(-(1 << 31), -32768)))       # 'i' in BAM
    if u == 8:
        return ("XZ", "Z", "RGZgrp" + "AB"[int(rng.integers(0, 2))])    # a value that LOOKS like the field
    return ("XB", "B", ("C", [82, 71, 90, 103, 114, 112, 65 + int(rng.integers(0, 2)), 0]))   # "RGZgrpA\0" as array bytes


def fuzz_dataset(seed: int, n_reads: int = 1500, contig_lens=(5000, 1200, 300), with_rg: bool = False,
                 extras: bool | None = None):
    """A genome + alignments that poke at every branch of both process_aln functions.
    Returns (contigs[(id, text)], refs[(name, len)] for the BAM header, recs).
    extras (default: with_rg): aux fields of every type around RG:Z, MAPQ 255, read names of 200+
    characters."""
    rng = np.random.default_rng(seed)
    if extras is None:
        extras = with_rg
    names = ["chrB", "chrA", "scaffold_10", "tiny.4", "tiny.5", "tiny.6"][:len(contig_lens)]
    contigs = [(nm, random_contig(rng, ln)) for nm, ln in zip(names, contig_lens)]
    refs = [(nm, len(s)) for nm, s in contigs] + [("chrMissing", 4000)]   # in BAM header, not in FASTA
    recs: list[Rec] = []
    for i in range(n_reads):
        ci = int(rng.integers(0, len(contigs)))
        cname, ctext = contigs[ci]
        clen = len(ctext)
        L = int(rng.integers(1, 90)) if rng.random() < 0.9 else int(rng.integers(90, 260))
        L = min(L, max(1, clen - 1))
        u = rng.random()
        if u < 0.80:
            s = int(rng.integers(0, max(1, clen - L + 1)))          # anywhere incl. the edges
        elif u < 0.90:
            s = int(rng.integers(0, 6))                               # hugging the left end
        else:
            s = max(0, clen - L - int(rng.integers(0, 6)))           # hugging the right end
        ref_slice = ctext[s:s + L].upper()
        seq = list(ref_slice.ljust(L, "A"))
        for j in range(L):                                            # substitutions, Ns, damage
            v = rng.random()
            if v < 0.03:
                seq[j] = "ACGT"[int(rng.integers(0, 4))]
            elif v < 0.035:
                seq[j] = "N"
            elif v < 0.037:
                seq[j] = "RYM="[int(rng.integers(0, 4))]
        seq = "".join(seq)
        flag = 0
        if rng.random() < 0.5:
            flag |= 0x10
        tlen = 0
        v = rng.random()
        if v < 0.30:                                                  # paired flavours
            flag |= 0x1
            if rng.random() < 0.8:
                flag |= 0x2
            if rng.random() < 0.1:
                flag |= 0x8
            w = rng.random()
            if w < 0.45:
                flag |= 0x40
            elif w < 0.9:
                flag |= 0x80
            elif w < 0.95:
                flag |= 0xC0
            if rng.random() < 0.5:
                flag |= 0x20
            t = rng.random()
            tlen = L if t < 0.35 else -L if t < 0.7 else 0 if t < 0.8 else int(rng.integers(-400, 400))
        for bit, pr in ((0x4, 0.01), (0x100, 0.02), (0x200, 0.02), (0x400, 0.03), (0x800, 0.02)):
            if rng.random() < pr:
                flag |= bit
        c = rng.random()
        if c < 0.72:
            cigar = [(L, "M")]
        elif c < 0.78 and L >= 4:
            a = int(rng.integers(1, L - 1)); cigar = [(a, "S"), (L - a, "M")]
        elif c < 0.82 and L >= 4:
            a = int(rng.integers(1, L - 1)); cigar = [(a, "M"), (L - a, "S")]
        elif c < 0.86 and L >= 6:
            a = int(rng.integers(1, L - 3)); cigar = [(a, "M"), (2, "I"), (L - a - 2, "M")]
        elif c < 0.90 and L >= 4:
            a = int(rng.integers(1, L - 1)); cigar = [(a, "M"), (3, "D"), (L - a, "M")]
        elif c < 0.93:
            cigar = [(L, "=")]
        elif c < 0.95 and L >= 2:
            a = int(rng.integers(1, L)); cigar = [(a, "M"), (L - a, "M")]
        elif c < 0.97:
            cigar = []
        elif c < 0.985:
            cigar = [(L + 1, "M")]
        else:
            cigar = [(L, "X")]
        rname = cname
        v = rng.random()
        if v < 0.02:
            rname = "chrMissing"
        elif v < 0.03:
            rname = "*"
        qual = "*" if rng.random() < 0.04 else "".join(chr(33 + int(q)) for q in rng.integers(2, 42, size=L))
        if rng.random() < 0.01:
            seq, qual = "*", "*"
        tags = []
        if with_rg:
            g = rng.random()
            if g < 0.45:
                tags.append(("RG", "Z", "grpA"))
            elif g < 0.85:
                tags.append(("RG", "Z", "grpB"))
            if rng.random() < 0.5:
                tags.insert(0 if rng.random() < 0.5 else len(tags), ("NM", "i", int(rng.integers(0, 400))))
            if rng.random() < 0.2:
                tags.insert(0, ("XA", "A", "q"))
        pos1 = s + 1 if rng.random() > 0.005 else 0
        qname, mapq = f"r{i:07d}", int(rng.integers(0, 61))
        if extras:
            for _ in range(int(rng.integers(0, 4))):                  # in front of / behind / between the fields so far
                tags.insert(int(rng.integers(0, len(tags) + 1)), random_aux(rng))
            if rng.random() < 0.05:
                mapq = 255                                            # "mapping quality not available"
            if rng.random() < 0.04:
                qname += "_" + "x" * int(rng.integers(192, 246))      # l_read_name up to 254 + NUL
        recs.append(Rec(qname=qname, flag=flag, rname=rname, pos=pos1,
                        mapq=mapq, cigar=cigar, tlen=tlen, seq=seq, qual=qual, tags=tags))
    return contigs, refs, recs


def ref_safe(recs: list[Rec], klen: int | None = None) -> list[Rec]:
    """Drops the records on which the reference itself has undefined behaviour
    (preconditions P3 / P4 in oracle/pss_oracle.c), for comparisons against oracle/_ref."""
    out = []
    for r in recs:
        seq_len = len(r.seq)          # '*' counts 1, like strlen("*")
        if (r.flag & 1) and r.cigar == [(abs(r.tlen), "M")] and abs(r.tlen) > seq_len:
            continue                  # P3: stale read bytes
        if klen is not None and r.pos - 1 < klen // 2:
            continue                  # P4: indexes in front of the contig
        out.append(r)
    return out


def random_pss_opts(rng: np.random.Generator) -> PssOpts:
    # sets with non-ACGT members (and different ones per side) exercise the "other base" classes of
    # the packed reference; lower case can never match (the genome is upper-cased at load)
    ctx_choices = ["ACGT", "ACGT", "ACGT", "CT", "G", "ACGTN", "TA", "N", "NR", "ACGTNRY", "acgt", "GY"]
    lo = int(rng.choice([0, 0, 10, 25]))
    hi = int(rng.choice([250000000, 250000000, 60, 120]))
    return PssOpts(region_len=int(rng.choice([1, 5, 8, 15, 16, 17, 25, 30, 31, 40, 70])), min_read_len=lo,
                   max_read_len=hi, min_mq=int(rng.choice([0, 0, 20, 37])),
                   up_ctx=str(rng.choice(ctx_choices)), down_ctx=str(rng.choice(ctx_choices)),
                   merged_only=bool(rng.random() < 0.3))


def random_fk_opts(rng: np.random.Generator) -> FkOpts:
    lo = int(rng.choice([0, 0, 10, 25]))
    hi = int(rng.choice([250000000, 250000000, 60, 120]))
    return FkOpts(klen=int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9])), min_mq=int(rng.choice([0, 0, 20, 37])),
                  min_read_len=lo, max_read_len=hi, merged_only=bool(rng.random() < 0.3))


# ----------------------------------------------------------------------------------
# a minimal, independent BAM reader (tests only; the product reader is C)
# ----------------------------------------------------------------------------------


def bgzf_inflate(data: bytes) -> bytes:
    out, i = [], 0
    while i < len(data):
        assert data[i:i + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block"
        xlen = struct.unpack_from("<H", data, i + 10)[0]
        bsize = None
        j = i + 12
        while j < i + 12 + xlen:
            si1, si2, slen = data[j], data[j + 1], struct.unpack_from("<H", data, j + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, j + 4)[0]
            j += 4 + slen
        assert bsize is not None
        cdata = data[i + 12 + xlen:i + bsize + 1 - 8]
        out.append(zlib.decompress(cdata, -15))
        i += bsize + 1
    return b"".join(out)


def read_bam(path: Path) -> tuple[list[tuple[str, int]], np.ndarray]:
    """-> (refs[(name, len)], raw alignment-record bytes as uint8 array)"""
    raw = bgzf_inflate(Path(path).read_bytes())
    assert raw[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", raw, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", raw, o)[0]
    o += 4
    refs = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", raw, o)[0]
        name = raw[o + 4:o + 4 + ln - 1].decode()
        refs.append((name, struct.unpack_from("<i", raw, o + 4 + ln)[0]))
        o += 8 + ln
    return refs, np.frombuffer(raw, dtype=np.uint8, offset=o).copy()


def raw_records(refs: list[tuple[str, int]], recs: list[Rec]) -> np.ndarray:
    idx = {n: i for i, (n, _) in enumerate(refs)}
    return np.frombuffer(b"".join(bam_record(r, idx) for r in recs), dtype=np.uint8).copy()


def loaded_contigs(contigs: list[tuple[str, str]]) -> list[tuple[str, np.ndarray]]:
    """FASTA text form -> the in-memory form init_genome produces (upper case bytes)"""
    return [(cid, np.frombuffer(seq.upper().encode(), dtype=np.uint8).copy()) for cid, seq in contigs]
