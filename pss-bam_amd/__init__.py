"""ctypes face of the MI355X tally engine (libpssbam_hip.so, include/pssbam_hip.h).

This module is thin plumbing for tests/, bench.py and __graft_entry__: it loads the
C-ABI shared library and marshals arguments.  There is no computation here and no CPU
fallback: if the library (or a gfx950 device) is missing, construction raises.

The directory name contains a hyphen, so import it through `load_pkg()` in
__graft_entry__.py (importlib by path) under the module name `pss_bam_amd`.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
LIB_HIP = PKG_DIR / "libpssbam_hip.so"
LIB_HOST = PKG_DIR / "libpssbam_host.so"
LIB_SYNTH = PKG_DIR / "libpssbam_synth.so"

TALLY_PSS, TALLY_KMER = 1, 2
KERNEL_AUTO, KERNEL_SIMPLE, KERNEL_TILED = 0, 1, 2
ST_NAMES = ["records", "rg_dropped", "parse_skip", "no_contig", "pss_ok", "pss_filtered", "kmer_ok",
            "kmer_filtered", "kmer_fail", "slow_path"]
ST_N = 16

# every symbol include/pssbam_hip.h declares (checked by tests/test_cabi.py)
HIP_SYMBOLS = [
    "pssbam_last_error", "pssbam_device_count", "pssbam_warmup", "pssbam_engine_create", "pssbam_engine_destroy",
    "pssbam_engine_set_stream", "pssbam_engine_set_genome", "pssbam_engine_set_genome_arrays",
    "pssbam_engine_set_references", "pssbam_engine_submit", "pssbam_engine_submit_async", "pssbam_engine_wait_copied",
    "pssbam_engine_copy_done", "pssbam_engine_phase_times", "pssbam_engine_submit_device", "pssbam_engine_sync",
    "pssbam_engine_finish", "pssbam_engine_reset", "pssbam_engine_counters_device", "pssbam_engine_bind_counters",
    "pssbam_reduce_counters", "pssbam_engine_genome_kmer_count", "pssbam_host_register", "pssbam_host_unregister", "pssbam_engine_timer_begin",
    "pssbam_engine_timer_end", "pssbam_engine_kernel_time", "pssbam_index_records", "pssbam_bgzf_scan",
    "pssbam_bgzf_inflate_device", "pssbam_bgzf_inflate_host", "pssbam_engine_submit_bgzf", "pssbam_engine_wait_bgzf_copied",
    "pssbam_engine_feed_status", "pssbam_engine_feed_break", "pssbam_engine_feed_handoff", "pssbam_feed_reserve", "pssbam_engine_hint_records",
    "pssbam_engine_set_genome_async", "pssbam_engine_genome_wait", "pssbam_engine_feed_open", "pssbam_feed_release",
]
EBUSY = -7


def build(verbose: bool = False) -> None:
    """make -C pss-bam_amd: compiles every HIP extension for gfx950 plus the host C side."""
    subprocess.run(["make", "-C", str(PKG_DIR), "all"], check=True,
                   stdout=None if verbose else subprocess.DEVNULL)


class PssbamError(RuntimeError):
    pass


class _PssOpts(C.Structure):
    _fields_ = [("region_len", C.c_int32), ("min_read_len", C.c_uint64), ("max_read_len", C.c_uint64),
                ("min_mq", C.c_int32), ("up_ctx", C.c_char_p), ("down_ctx", C.c_char_p),
                ("merged_only", C.c_int32)]


class _KmerOpts(C.Structure):
    _fields_ = [("klen", C.c_int32), ("min_mq", C.c_int32), ("min_read_len", C.c_uint64),
                ("max_read_len", C.c_uint64), ("merged_only", C.c_int32)]


class _Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("tally_mask", C.c_uint32), ("pss", _PssOpts), ("kmer", _KmerOpts),
                ("read_group", C.c_char_p), ("device", C.c_int32), ("kernel", C.c_int32)]


_hip = None


def hip_lib() -> C.CDLL:
    """Loads libpssbam_hip.so; raises (loudly) when it has not been built."""
    global _hip
    if _hip is not None:
        return _hip
    if not LIB_HIP.exists():
        raise PssbamError(f"{LIB_HIP} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    L = C.CDLL(str(LIB_HIP))
    L.pssbam_last_error.restype = C.c_char_p
    L.pssbam_engine_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    L.pssbam_engine_destroy.argtypes = [C.c_void_p]
    L.pssbam_engine_destroy.restype = None
    L.pssbam_engine_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.pssbam_engine_set_genome.argtypes = [C.c_void_p, C.c_void_p]
    L.pssbam_engine_set_genome_arrays.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p),
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int]
    L.pssbam_engine_set_references.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_char_p)]
    L.pssbam_engine_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
    L.pssbam_engine_submit_async.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    L.pssbam_engine_wait_copied.argtypes = [C.c_void_p, C.c_uint64]
    L.pssbam_engine_copy_done.argtypes = [C.c_void_p, C.c_uint64]
    L.pssbam_engine_phase_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_uint64)]
    L.pssbam_engine_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
    L.pssbam_engine_sync.argtypes = [C.c_void_p]
    L.pssbam_engine_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.pssbam_engine_reset.argtypes = [C.c_void_p]
    L.pssbam_engine_counters_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.pssbam_engine_bind_counters.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.pssbam_engine_genome_kmer_count.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.pssbam_engine_timer_begin.argtypes = [C.c_void_p]
    L.pssbam_engine_timer_end.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.pssbam_engine_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
    L.pssbam_index_records.restype = C.c_int64
    L.pssbam_index_records.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    _hip = L
    return L


def _chk(rc: int) -> None:
    if rc != 0:
        raise PssbamError(f"pssbam error {rc}: {hip_lib().pssbam_last_error().decode()}")


def bgzf_inflate(bgzf: np.ndarray, check_crc: bool = True, repeats: int = 1, want_output: bool = True) -> dict:
    """Inflates whole BGZF blocks on the GPU (pssbam_bgzf_inflate_host): {"data", "n_blocks", "bad_block",
    "bad_status", "kernel_ms"}; bad_block is None when every block passed."""
    L = hip_lib()
    L.pssbam_bgzf_inflate_host.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_double), C.c_int, C.c_int]
    L.pssbam_bgzf_scan.restype = C.c_int64
    L.pssbam_bgzf_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    bgzf = np.ascontiguousarray(bgzf, dtype=np.uint8)
    consumed, total = C.c_uint64(), C.c_uint64()
    n = L.pssbam_bgzf_scan(bgzf.ctypes.data, bgzf.size, None, 0, C.byref(consumed), C.byref(total))
    if n < 0:
        _chk(int(n))
    out = np.empty(int(total.value) if want_output else 0, dtype=np.uint8)
    out_len, nb, bad_b, bad_s, ms = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_double()
    _chk(L.pssbam_bgzf_inflate_host(-1, bgzf.ctypes.data, bgzf.size, out.ctypes.data if want_output else None, out.size,
                                    C.byref(out_len), C.byref(nb), C.byref(bad_b), C.byref(bad_s), C.byref(ms),
                                    int(check_crc), repeats))
    return {"data": out, "inflated_bytes": int(out_len.value), "n_blocks": int(nb.value),
            "bad_block": None if bad_b.value == 0xFFFFFFFF else int(bad_b.value), "bad_status": int(bad_s.value),
            "kernel_ms": float(ms.value)}


def index_records(buf: np.ndarray) -> np.ndarray:
    """offsets (n+1,) u32 of the whole records in an inflated BAM record stream"""
    L = hip_lib()
    n = L.pssbam_index_records(buf.ctypes.data, buf.size, None, 1 << 62, None)
    if n < 0:
        _chk(int(n))
    offs = np.empty(n + 1, dtype=np.uint32)
    consumed = C.c_uint64()
    n2 = L.pssbam_index_records(buf.ctypes.data, buf.size, offs.ctypes.data, n, C.byref(consumed))
    assert n2 == n
    return offs


@dataclass
class Tables:
    fwd: np.ndarray | None
    rev: np.ndarray | None
    k5: np.ndarray | None
    k3: np.ndarray | None
    stats: dict


class Engine:
    """One engine = one GPU, one stream.  Options mirror the two reference CLIs:
    `pss` = dict(region_len, min_read_len, max_read_len, min_mq, up_ctx, down_ctx, merged_only),
    `kmer` = dict(klen, min_mq, min_read_len, max_read_len, merged_only)."""

    def __init__(self, pss: dict | None = None, kmer: dict | None = None, read_group: str | None = None,
                 kernel: int = KERNEL_AUTO, device: int = -1):
        L = hip_lib()
        cfg = _Config()
        cfg.abi_version = 1
        cfg.tally_mask = (TALLY_PSS if pss is not None else 0) | (TALLY_KMER if kmer is not None else 0)
        self._keep = []
        if pss is not None:
            up, dn = pss.get("up_ctx", "ACGT").encode(), pss.get("down_ctx", "ACGT").encode()
            self._keep += [up, dn]
            cfg.pss = _PssOpts(pss.get("region_len", 15), pss.get("min_read_len", 0),
                               pss.get("max_read_len", 250000000), pss.get("min_mq", 0), up, dn,
                               int(pss.get("merged_only", False)))
            self.region_len = cfg.pss.region_len
        if kmer is not None:
            cfg.kmer = _KmerOpts(kmer.get("klen", 8), kmer.get("min_mq", 0), kmer.get("min_read_len", 0),
                                 kmer.get("max_read_len", 250000000), int(kmer.get("merged_only", False)))
            self.klen = cfg.kmer.klen
        self.has_pss, self.has_kmer = pss is not None, kmer is not None
        if read_group is not None:
            rg = read_group.encode()
            self._keep.append(rg)
            cfg.read_group = rg
        cfg.device = device
        cfg.kernel = kernel
        h = C.c_void_p()
        _chk(L.pssbam_engine_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.pssbam_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def set_stream(self, hip_stream: int):
        _chk(self._L.pssbam_engine_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_genome_arrays(self, contigs: list[tuple[str, np.ndarray]]):
        """contigs: [(id, uint8 array in loaded form)] (host memory)"""
        n = len(contigs)
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for _, a in contigs]
        ids = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
        seqs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        lens = (C.c_uint64 * n)(*[a.size for a in arrs])
        _chk(self._L.pssbam_engine_set_genome_arrays(self._h, n, ids, seqs, lens, 0))

    def set_genome_device(self, contigs: list[tuple[str, int, int]]):
        """contigs: [(id, device_ptr, length)]"""
        n = len(contigs)
        ids = (C.c_char_p * n)(*[c[0].encode() for c in contigs])
        seqs = (C.c_void_p * n)(*[c[1] for c in contigs])
        lens = (C.c_uint64 * n)(*[c[2] for c in contigs])
        _chk(self._L.pssbam_engine_set_genome_arrays(self._h, n, ids, seqs, lens, 1))

    def set_genome_struct(self, genome_ptr: int):
        """genome_ptr: a Genome* from init_genome (libpssbam_host.so)"""
        _chk(self._L.pssbam_engine_set_genome(self._h, C.c_void_p(genome_ptr)))

    def set_references(self, names: list[str]):
        n = len(names)
        arr = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
        _chk(self._L.pssbam_engine_set_references(self._h, n, arr))

    def submit(self, records: np.ndarray, offsets: np.ndarray | None = None):
        records = np.ascontiguousarray(records, dtype=np.uint8)
        if offsets is None:
            offsets = index_records(records)
            if int(offsets[-1]) != records.size:
                raise PssbamError("record block ends in a partial record")
        offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
        _chk(self._L.pssbam_engine_submit(self._h, records.ctypes.data, records.size, offsets.ctypes.data,
                                          offsets.size - 1))

    def submit_async(self, records: np.ndarray, offsets: np.ndarray) -> int:
        """enqueue only; the arrays must stay alive and untouched until wait_copied(ticket)"""
        t = C.c_uint64()
        _chk(self._L.pssbam_engine_submit_async(self._h, records.ctypes.data, records.size, offsets.ctypes.data,
                                                offsets.size - 1, C.byref(t)))
        return int(t.value)

    def wait_copied(self, ticket: int):
        _chk(self._L.pssbam_engine_wait_copied(self._h, ticket))

    def copy_done(self, ticket: int) -> bool:
        rc = self._L.pssbam_engine_copy_done(self._h, ticket)
        if rc < 0:
            _chk(rc)
        return bool(rc)

    def phase_times(self) -> dict:
        a, b, c, d = C.c_double(), C.c_uint64(), C.c_double(), C.c_uint64()
        _chk(self._L.pssbam_engine_phase_times(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return {"h2d_ms": a.value, "h2d_bytes": b.value, "kernel_ms": c.value, "launches": d.value}

    def feed_open(self, n_ref: int, genome_bytes_hint: int = 0):
        """compressed blocks may be fed before set_genome / set_references; their tallies follow then"""
        self._L.pssbam_engine_feed_open.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
        _chk(self._L.pssbam_engine_feed_open(self._h, n_ref, genome_bytes_hint))

    def submit_bgzf(self, bgzf: np.ndarray, header_bytes: int = 0, max_batch_inflated: int = 1 << 30, on_busy=None) -> int:
        """Whole BGZF blocks (host bytes) through the device-side feed: inflate, CRC, record index and
        tally on the GPU.  header_bytes = inflated bytes in front of the first alignment record (the BAM
        header when `bgzf` starts at the beginning of the file).  Returns the number of blocks.
        on_busy: called when the engine answers PSSBAM_EBUSY (fed ahead of the genome, every slot full); it
        must set the genome and references, after which the chunk is submitted again."""
        class _Blk(C.Structure):
            _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint32), ("isize", C.c_uint32), ("out_off", C.c_uint64),
                        ("crc", C.c_uint32), ("status", C.c_uint32)]
        L = self._L
        L.pssbam_bgzf_scan.restype = C.c_int64
        L.pssbam_bgzf_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.pssbam_engine_submit_bgzf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32,
                                                C.POINTER(C.c_uint64)]
        L.pssbam_engine_wait_bgzf_copied.argtypes = [C.c_void_p, C.c_uint64]
        bgzf = np.ascontiguousarray(bgzf, dtype=np.uint8)
        consumed = C.c_uint64()
        n = L.pssbam_bgzf_scan(bgzf.ctypes.data, bgzf.size, None, 0, C.byref(consumed), None)
        if n < 0:
            _chk(int(n))
        if consumed.value != bgzf.size:
            raise PssbamError("input ends inside a BGZF block")
        blocks = (_Blk * max(int(n), 1))()
        L.pssbam_bgzf_scan(bgzf.ctypes.data, bgzf.size, blocks, n, None, None)
        i, skip = 0, header_bytes
        while i < n and skip >= blocks[i].isize and (skip > 0 or blocks[i].isize == 0):   # blocks that are all header
            skip -= blocks[i].isize
            i += 1
        while i < n:
            j, base_in, base_out = i, blocks[i].in_off & ~3, blocks[i].out_off
            while j < n and blocks[j].out_off + blocks[j].isize - base_out <= max_batch_inflated:
                j += 1
            j = max(j, i + 1)
            grp = (_Blk * (j - i))()
            for k in range(i, j):
                grp[k - i] = blocks[k]
                grp[k - i].in_off -= base_in
                grp[k - i].out_off -= base_out
            end_in = blocks[j - 1].in_off + blocks[j - 1].in_len
            t = C.c_uint64()
            rc = L.pssbam_engine_submit_bgzf(self._h, bgzf.ctypes.data + base_in, end_in - base_in, grp, j - i, skip, C.byref(t))
            if rc == EBUSY and on_busy is not None:
                on_busy()
                rc = L.pssbam_engine_submit_bgzf(self._h, bgzf.ctypes.data + base_in, end_in - base_in, grp, j - i, skip, C.byref(t))
            _chk(rc)
            _chk(L.pssbam_engine_wait_bgzf_copied(self._h, t.value))
            skip = 0
            i = j
        return int(n)

    def feed_break(self):
        self._L.pssbam_engine_feed_break.argtypes = [C.c_void_p]
        _chk(self._L.pssbam_engine_feed_break(self._h))

    def feed_handoff(self, to: "Engine"):
        """the blocks submitted to `to` from now on continue this engine's stream (include/pssbam_hip.h)"""
        self._L.pssbam_engine_feed_handoff.argtypes = [C.c_void_p, C.c_void_p]
        _chk(self._L.pssbam_engine_feed_handoff(self._h, to._h))

    def feed_status(self) -> dict:
        L = self._L
        L.pssbam_engine_feed_status.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        f, ms, nb = C.c_uint32(), C.c_double(), C.c_uint64()
        _chk(L.pssbam_engine_feed_status(self._h, C.byref(f), C.byref(ms), C.byref(nb)))
        return {"flags": int(f.value), "inflate_ms": float(ms.value), "inflated_bytes": int(nb.value)}

    def submit_device(self, d_records: int, nbytes: int, d_offsets: int, n_records: int):
        _chk(self._L.pssbam_engine_submit_device(self._h, C.c_void_p(d_records), nbytes, C.c_void_p(d_offsets),
                                                 n_records))

    def sync(self):
        _chk(self._L.pssbam_engine_sync(self._h))

    def reset(self):
        _chk(self._L.pssbam_engine_reset(self._h))

    def finish(self) -> Tables:
        fwd = rev = k5 = k3 = None
        if self.has_pss:
            fwd = np.zeros((self.region_len + 2, 16), dtype=np.uint64)
            rev = np.zeros_like(fwd)
        if self.has_kmer:
            k5 = np.zeros(4 ** self.klen, dtype=np.uint64)
            k3 = np.zeros_like(k5)
        st = np.zeros(ST_N, dtype=np.uint64)
        p = lambda a: a.ctypes.data if a is not None else None  # noqa: E731
        _chk(self._L.pssbam_engine_finish(self._h, p(fwd), p(rev), p(k5), p(k3), st.ctypes.data))
        return Tables(fwd, rev, k5, k3, {n: int(st[i]) for i, n in enumerate(ST_NAMES)})

    def counters_device(self) -> tuple[int, int]:
        ptr, n = C.c_void_p(), C.c_size_t()
        _chk(self._L.pssbam_engine_counters_device(self._h, C.byref(ptr), C.byref(n)))
        return int(ptr.value), int(n.value)

    def bind_counters(self, d_ptr: int | None, n_u64: int = 0):
        _chk(self._L.pssbam_engine_bind_counters(self._h, C.c_void_p(d_ptr) if d_ptr else None, n_u64))

    def counter_layout(self) -> dict:
        """u64-word offsets of the sections of the counter block"""
        rows = (self.region_len + 2) if self.has_pss else 0
        nb = 4 ** self.klen if self.has_kmer else 0
        return {"fwd": 0, "rev": rows * 16, "k5": 2 * rows * 16, "k3": 2 * rows * 16 + nb,
                "stats": 2 * rows * 16 + 2 * nb, "rows": rows, "bins": nb}

    def genome_kmer_count(self, klen: int) -> np.ndarray:
        out = np.zeros(4 ** klen, dtype=np.uint64)
        _chk(self._L.pssbam_engine_genome_kmer_count(self._h, klen, out.ctypes.data))
        return out

    def timer_begin(self):
        _chk(self._L.pssbam_engine_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        _chk(self._L.pssbam_engine_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    def kernel_time(self, reset: bool = True) -> tuple[float, int]:
        ms, n = C.c_double(), C.c_uint64()
        _chk(self._L.pssbam_engine_kernel_time(self._h, C.byref(ms), C.byref(n), int(reset)))
        return float(ms.value), int(n.value)
