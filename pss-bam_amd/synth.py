"""ctypes face of libpssbam_synth.so -- the synthetic workloads of SURVEY 8d (bench / test
infrastructure).  See csrc/synth_model.h for the model itself."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
LIB = PKG_DIR / "libpssbam_synth.so"
MAXC = 32

# hg19-like contig lengths scaled so the total is ~3.0 Gb (chr1..22, X, Y), each < 536870911
HG19_LENS = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431,
             135534747, 135006516, 133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248,
             59128983, 63025520, 48129895, 51304566, 155270560, 59373566]


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_contigs", C.c_uint32), ("name_mode", C.c_uint32),
                ("contig_len", C.c_uint64 * MAXC), ("n_reads", C.c_uint64), ("len_min", C.c_uint32),
                ("len_max", C.c_uint32), ("sorted", C.c_uint32), ("cigar_mix", C.c_uint32), ("damage", C.c_uint32),
                ("sub_per_64k", C.c_uint32), ("dup_per_1k", C.c_uint32), ("lowmq_per_1k", C.c_uint32),
                ("n_run_len", C.c_uint32), ("pad_", C.c_uint32), ("usable_first", C.c_uint64 * (MAXC + 1)),
                ("perm_mul", C.c_uint64), ("perm_add", C.c_uint64)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB.exists():
            raise RuntimeError(f"{LIB} missing: run __graft_entry__.build()")
        L = C.CDLL(str(LIB))
        P = C.POINTER(SynthCfg)
        L.synth_cfg_finish.argtypes = [P]
        L.synth_contig_name.argtypes = [P, C.c_uint32, C.c_char_p]
        L.synth_genome_host.argtypes = [P, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int]
        L.synth_genome_device.argtypes = [P, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.synth_sizes_host.argtypes = [P, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.synth_records_host.argtypes = [P, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
        L.synth_records_device.argtypes = [P, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.synth_offsets_linear_device.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
        L.synth_sam_host.argtypes = [P, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
        L.synth_fasta_host.argtypes = [P, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.synth_bam_file_host.argtypes = [P, C.c_uint64, C.c_uint64, C.c_char_p, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def make_cfg(seed: int, contig_lens: list[int], n_reads: int, len_min: int, len_max: int, sorted_: bool = True,
             cigar_mix: bool = False, damage: bool = False, sub_rate: float = 0.01, dup_per_1k: int = 0,
             lowmq_per_1k: int = 0, n_run_len: int = 41, name_mode: int = 1) -> SynthCfg:
    c = SynthCfg()
    c.seed = seed
    c.n_contigs = len(contig_lens)
    c.name_mode = name_mode
    for i, ln in enumerate(contig_lens):
        c.contig_len[i] = ln
    c.n_reads = n_reads
    c.len_min, c.len_max = len_min, len_max
    c.sorted = int(sorted_)
    c.cigar_mix, c.damage = int(cigar_mix), int(damage)
    c.sub_per_64k = int(round(sub_rate * 65536))
    c.dup_per_1k, c.lowmq_per_1k = dup_per_1k, lowmq_per_1k
    c.n_run_len = n_run_len
    if lib().synth_cfg_finish(C.byref(c)) != 0:
        raise ValueError("bad synthetic configuration")
    return c


# the named configurations of BASELINE.json / SURVEY 8d
def config(name: str, n_reads: int | None = None, sorted_: bool = True, scale_genome: float = 1.0) -> dict:
    lens = [max(2000, int(x * scale_genome)) for x in HG19_LENS]
    table = {
        "C1": dict(seed=1, contig_lens=[1_000_000], n_reads=1_000_000, len_min=100, len_max=100, sorted_=False,
                   name_mode=0, n_run_len=0, region_len=15),
        "C2": dict(seed=2, contig_lens=lens, n_reads=50_000_000, len_min=150, len_max=150, region_len=25),
        "C3": dict(seed=3, contig_lens=lens, n_reads=200_000_000, len_min=150, len_max=150, region_len=25),
        "C4": dict(seed=4, contig_lens=lens, n_reads=100_000_000, len_min=30, len_max=80, cigar_mix=True, damage=True,
                   dup_per_1k=20, lowmq_per_1k=20, region_len=15),
        "C5": dict(seed=5, contig_lens=lens, n_reads=200_000_000, len_min=150, len_max=150, region_len=25, klen=4),
    }
    d = dict(table[name])
    if n_reads is not None:
        d["n_reads"] = n_reads
    if name != "C1":
        d["sorted_"] = sorted_
    return d


def contig_name(cfg: SynthCfg, k: int) -> str:
    buf = C.create_string_buffer(16)
    lib().synth_contig_name(C.byref(cfg), k, buf)
    return buf.value.decode()


def genome_host(cfg: SynthCfg, contig: int, p0: int = 0, n: int | None = None, fasta_case: bool = False,
                threads: int = 8) -> np.ndarray:
    n = int(cfg.contig_len[contig]) - p0 if n is None else n
    out = np.empty(n, dtype=np.uint8)
    lib().synth_genome_host(C.byref(cfg), contig, out.ctypes.data, p0, n, int(fasta_case), threads)
    return out


def sizes_host(cfg: SynthCfg, slot0: int, n: int, threads: int = 8) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().synth_sizes_host(C.byref(cfg), slot0, n, out.ctypes.data, threads)
    return out


def records_host(cfg: SynthCfg, slot0: int, n: int, threads: int = 8) -> tuple[np.ndarray, np.ndarray]:
    """-> (record bytes, offsets[n+1])"""
    sizes = sizes_host(cfg, slot0, n, threads)
    offs = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(sizes, out=offs[1:])
    assert int(offs[-1]) < 2 ** 32
    offs32 = offs.astype(np.uint32)
    out = np.empty(int(offs[-1]), dtype=np.uint8)
    lib().synth_records_host(C.byref(cfg), slot0, n, offs32.ctypes.data, out.ctypes.data, threads)
    return out, offs32


def sam_host(cfg: SynthCfg, slot0: int, n: int, path, with_header: bool = True) -> None:
    if lib().synth_sam_host(C.byref(cfg), slot0, n, str(path).encode(), int(with_header)) != 0:
        raise OSError(f"cannot write {path}")


def fasta_host(cfg: SynthCfg, path, first: int = 0, count: int | None = None, width: int = 60, threads: int = 8):
    count = int(cfg.n_contigs) - first if count is None else count
    if lib().synth_fasta_host(C.byref(cfg), str(path).encode(), first, count, width, threads) != 0:
        raise OSError(f"cannot write {path}")


def bam_file_host(cfg: SynthCfg, slot0: int, n: int, path, level: int = 1, threads: int = 8, ragged: bool = False,
                  quals: str = "const") -> None:
    """BGZF-compressed BAM file (header + records of slots [slot0, slot0+n)).  Default block layout is
    htslib's (whole records per block); ragged=True cuts the stream every 0xff00 bytes instead.
    quals: "const" (the named configurations: QUAL all 'I'), "binned" (four quality bins, i.i.d. per base)
    or "full" (40 levels) -- only the compressed size and the inflate's work change, never the tables."""
    layout = int(ragged) | {"const": 0, "binned": 2, "full": 4}[quals]
    if lib().synth_bam_file_host(C.byref(cfg), slot0, n, str(path).encode(), level, threads, layout) != 0:
        raise OSError(f"cannot write {path}")
