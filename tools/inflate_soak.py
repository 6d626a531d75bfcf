#!/usr/bin/env python3
"""tools/inflate_soak.py -- GPU-box helper: many more damaged and hostile BGZF blocks through both data loops of the
inflate kernel than the test-suite carries (tests/test_gpu_inflate.py).  Every call must return; a block is either
flagged or inflates to what zlib makes of it.
    python3 tools/inflate_soak.py [--trials 400] [--hostile 20000]"""
import argparse
import os
import struct
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=400)
ap.add_argument("--hostile", type=int, default=20000)
args = ap.parse_args()
pkg = ge.load_pkg()


def bgzf(data: bytes, level=6) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = c.compress(data) + c.flush()
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(payload) + 25) + payload
            + struct.pack("<II", zlib.crc32(data), len(data)))


def zlib_inflate(block: bytes):
    try:
        d = zlib.decompressobj(-15)
        out = d.decompress(block[18:-8]) + d.flush()
        return out if d.eof else None
    except zlib.error:
        return None


rng = np.random.default_rng(2026)
for loop, pieces in (("0", "1"), ("1", "1"), ("1", "0")):   # in place / one wait with bounded pieces (the default) / one wait, whole copies
    os.environ["PSSBAM_INFLATE_LOOP"] = loop
    os.environ["PSSBAM_INFLATE_PIECES"] = pieces
    flagged = passed = 0
    # (a) valid blocks of mixed content with random bit flips / byte splices anywhere in the payload
    for trial in range(args.trials):
        n_blk = 64
        datas = []
        for i in range(n_blk):
            kind = int(rng.integers(0, 4))
            n = int(rng.integers(1, 65000))
            if kind == 0:
                d = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            elif kind == 1:
                d = bytes(rng.integers(65, 69, n, dtype=np.uint8))
            elif kind == 2:
                d = (b"read%07d\tACGTTGCA" % i) * (n // 20 + 1)
                d = d[:n]
            else:
                d = bytes([int(rng.integers(0, 256))]) * n
            datas.append(d)
        blocks = [bytearray(bgzf(d, int(rng.integers(1, 10)))) for d in datas]
        hit = set()
        for _ in range(int(rng.integers(1, 12))):
            k = int(rng.integers(0, n_blk))
            b = blocks[k]
            if len(b) <= 27:
                continue
            o = int(rng.integers(18, len(b) - 8))
            if rng.integers(0, 2):
                b[o] ^= 1 << int(rng.integers(0, 8))
            else:
                m = min(int(rng.integers(1, 9)), len(b) - 8 - o)
                b[o:o + m] = bytes(rng.integers(0, 256, m, dtype=np.uint8))
            hit.add(k)
        buf = b"".join(bytes(b) for b in blocks)
        res = pkg.bgzf_inflate(np.frombuffer(buf, dtype=np.uint8))
        out = res["data"].tobytes()
        # every untouched block must come out right; the first bad block, if any, must be a touched one
        if res["bad_block"] is not None:
            assert res["bad_block"] in hit, (loop, trial, res["bad_block"], sorted(hit))
            flagged += 1
        else:
            passed += 1
        off = 0
        for k, d in enumerate(datas):
            if k not in hit:
                assert out[off:off + len(d)] == d, (loop, trial, k)
            off += len(d)
    # (b) random bytes dressed up as blocks
    blocks = []
    for i in range(args.hostile):
        n = int(rng.integers(1, 4000))
        payload = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        if i % 3 == 0:
            payload = bytes([0b101]) + payload
        elif i % 3 == 1:
            payload = bytes([0b011]) + payload
        isize = int(rng.integers(0, 65537))
        blocks.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(payload) + 25) + payload
                      + struct.pack("<II", int(rng.integers(0, 1 << 32)), isize))
    res = pkg.bgzf_inflate(np.frombuffer(b"".join(blocks), dtype=np.uint8))
    assert res["n_blocks"] == args.hostile
    print(f"loop {loop} pieces {pieces}: {args.trials} mutated files ({flagged} flagged, {passed} harmless), {args.hostile} hostile blocks: returned, first bad block {res['bad_block']}", flush=True)
print("soak ok")
