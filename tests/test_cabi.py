"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads,
exports every symbol include/pssbam_hip.h declares, and refuses to run without a GPU
(no CPU fallback exists).  No compute is attempted here."""
import re
from pathlib import Path

import numpy as np
import pytest

import __graft_entry__ as ge

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def pkg():
    ge.build()
    return ge.load_pkg()


def test_header_symbols_are_exported(pkg):
    hdr = (ROOT / "include" / "pssbam_hip.h").read_text()
    declared = set(re.findall(r"\b(pssbam_[a-z_]+)\s*\(", hdr))
    declared -= {"pssbam_engine"}
    assert declared == set(pkg.HIP_SYMBOLS), declared ^ set(pkg.HIP_SYMBOLS)
    lib = pkg.hip_lib()
    for s in declared:
        assert hasattr(lib, s), s


def test_code_object_is_gfx950_only(pkg):
    """the fat binary inside the shared library carries gfx950 code and nothing else"""
    blob = pkg.LIB_HIP.read_bytes()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_index_records_host_helper(pkg):
    import pssbam_testlib as tl
    contigs, refs, recs = tl.fuzz_dataset(3, 200)
    raw = tl.raw_records(refs, recs)
    offs = pkg.index_records(raw)
    assert offs.size == len(recs) + 1 and offs[0] == 0 and int(offs[-1]) == raw.size
    # a trailing partial record is left for the caller
    offs2 = pkg.index_records(raw[:-5])
    assert offs2.size == len(recs) and int(offs2[-1]) == int(offs[-2])
    # block_size < 32 is malformed
    bad = raw.copy()
    bad[0:4] = np.frombuffer((7).to_bytes(4, "little"), dtype=np.uint8)
    with pytest.raises(pkg.PssbamError):
        pkg.index_records(bad)


def test_engine_refuses_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PssbamError) as ei:
        pkg.Engine(pss=dict(region_len=15))
    assert "no HIP device" in str(ei.value) or "-2" in str(ei.value)


def test_bgzf_scan_host_helper(pkg):
    """pssbam_bgzf_scan (host code, no GPU): block table of a BGZF file -- payload offsets/lengths,
    ISIZE, CRC, running output offsets -- checked against zlib block by block; partial trailing
    block left to the caller; non-BGZF bytes refused"""
    import ctypes as C
    import zlib
    import pssbam_testlib as tl

    class Blk(C.Structure):
        _fields_ = [("in_off", C.c_uint64), ("in_len", C.c_uint32), ("isize", C.c_uint32), ("out_off", C.c_uint64),
                    ("crc", C.c_uint32), ("status", C.c_uint32)]
    L = pkg.hip_lib()
    L.pssbam_bgzf_scan.restype = C.c_int64
    L.pssbam_bgzf_scan.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    raw = (ROOT / "tests" / "golden" / "setA.bam").read_bytes()
    buf = np.frombuffer(raw, dtype=np.uint8)
    consumed, total = C.c_uint64(), C.c_uint64()
    n = L.pssbam_bgzf_scan(buf.ctypes.data, buf.size, None, 0, C.byref(consumed), C.byref(total))
    assert n >= 2 and consumed.value == len(raw) and total.value == len(tl.bgzf_inflate(raw))
    blocks = (Blk * n)()
    assert L.pssbam_bgzf_scan(buf.ctypes.data, buf.size, blocks, n, None, None) == n
    off = 0
    for b in blocks:
        data = zlib.decompress(raw[b.in_off:b.in_off + b.in_len], -15)
        assert len(data) == b.isize and (zlib.crc32(data) & 0xFFFFFFFF) == b.crc and b.out_off == off
        off += b.isize
    # a cut-off last block is not counted, and its bytes are not consumed
    m = L.pssbam_bgzf_scan(buf.ctypes.data, buf.size - 5, None, 0, C.byref(consumed), None)
    assert m == n - 1 and consumed.value == len(raw) - 28      # (the last block is the 28-byte EOF marker)
    junk = np.frombuffer(b"this is not a BGZF file at all, just some text" * 3, dtype=np.uint8)
    assert L.pssbam_bgzf_scan(junk.ctypes.data, junk.size, None, 0, None, None) < 0


def test_null_engines_are_refused_not_dereferenced(pkg):
    """error convention of the boundary (include/pssbam_hip.h): a null engine is PSSBAM_EINVAL with a message, on every
    entry point of the compressed feed -- checked here for the ones that touch no GPU before they look at their arguments"""
    import ctypes as C
    L = pkg.hip_lib()
    L.pssbam_last_error.restype = C.c_char_p
    calls = [("pssbam_engine_feed_handoff", [C.c_void_p, C.c_void_p], (None, None)),
             ("pssbam_engine_feed_break", [C.c_void_p], (None,)),
             ("pssbam_engine_feed_open", [C.c_void_p, C.c_int32, C.c_uint64], (None, 3, 0)),
             ("pssbam_engine_submit_bgzf", [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p], (None, None, 0, None, 0, 0, None)),
             ("pssbam_engine_feed_status", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p], (None, None, None, None)),
             ("pssbam_engine_set_genome_async", [C.c_void_p, C.c_void_p], (None, None)),
             ("pssbam_engine_genome_wait", [C.c_void_p], (None,))]
    for name, argtypes, args in calls:
        f = getattr(L, name)
        f.argtypes = argtypes
        f.restype = C.c_int
        assert f(*args) == -1, name                  # PSSBAM_EINVAL
        assert L.pssbam_last_error(), name
