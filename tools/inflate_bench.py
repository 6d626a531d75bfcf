#!/usr/bin/env python3
"""tools/inflate_bench.py -- GPU-box helper: device-side BGZF inflate (csrc/inflate_kernels.h) on a
generated BAM of the benchmark shape: payload GB/s of the inflate (+ CRC) kernels (HIP events, best of
3), bit-exactness against the host decoder, and the host reader's own inflate rate on the same file.
    python3 tools/inflate_bench.py [--reads 20000000] [--level 1] [--config C3] [--ragged]"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=20_000_000)
ap.add_argument("--level", type=int, default=1)
ap.add_argument("--config", default="C3")
ap.add_argument("--ragged", action="store_true")
ap.add_argument("--quals", default="const", choices=["const", "binned", "full"], help="QUAL model of the generated BAM (synth.bam_file_host)")
ap.add_argument("--no-crc", action="store_true")
ap.add_argument("--repeats", type=int, default=3, help="kernel runs; the best is reported (1 = a cold first run, what a short-lived command sees)")
ap.add_argument("--no-output", action="store_true", help="skip the D2H copy and the zlib comparison (large files)")
ap.add_argument("--verify-blocks", type=int, default=4000, help="blocks compared byte for byte with zlib (all are CRC-checked on the device)")
args = ap.parse_args()

pkg = ge.load_pkg()
from pss_bam_amd import synth  # noqa: E402

threads = bench.worker_threads()
d = synth.config(args.config, scale_genome=0.1)
d.pop("region_len")
d.pop("klen", None)
d["n_reads"] = 200_000_000 if args.config in ("C3", "C5") else d["n_reads"]
cfg = synth.make_cfg(**d)
tmp = Path(tempfile.mkdtemp(prefix="pssbam_inf_", dir=os.environ.get("TMPDIR", "/tmp")))
bam = tmp / "reads.bam"
t = time.time()
synth.bam_file_host(cfg, 0, args.reads, bam, level=args.level, threads=threads, ragged=args.ragged, quals=args.quals)
t_gen = time.time() - t
raw = np.fromfile(bam, dtype=np.uint8)
res = pkg.bgzf_inflate(raw, check_crc=not args.no_crc, repeats=args.repeats, want_output=not args.no_output)
assert res["bad_block"] is None, res
inflated = res["inflated_bytes"]
# byte-for-byte against zlib on a sample of blocks spread over the file
import struct
o, k, checked, uo = 0, 0, 0, 0
step = max(1, res["n_blocks"] // max(args.verify_blocks, 1))
data = res["data"]
buf = raw.tobytes()
while o < len(buf) and not args.no_output:
    bsize = struct.unpack_from("<H", buf, o + 16)[0] + 1
    isize = struct.unpack_from("<I", buf, o + bsize - 4)[0]
    if k % step == 0:
        want = zlib.decompress(buf[o + 18:o + bsize - 8], -15)
        assert data[uo:uo + isize].tobytes() == want, f"block {k} differs from zlib"
        checked += 1
    uo += isize
    o += bsize
    k += 1
# the host reader on the same file (its inflate stage only, PSSBAM_STATS wording of bin/hostcheck is not available: time bam2sam-free path)
t = time.time()
pr = subprocess.run([str(pkg.PKG_DIR / "bin" / "hostcheck"), "-q", "-a", str(bam)], capture_output=True, text=True)
t_host = time.time() - t
m = re.search(r"reader inflate stage ([\d.]+) s", pr.stderr)
t_host_inflate = float(m.group(1)) if m else None
out = {
    "reads": args.reads, "deflate_level": args.level, "layout": "ragged" if args.ragged else "htslib",
    "bam_bytes": int(raw.size), "inflated_bytes": int(inflated), "n_blocks": res["n_blocks"],
    "device_kernel_ms": res["kernel_ms"], "kernel_runs_best_of": args.repeats, "device_GBps_inflated": inflated / res["kernel_ms"] / 1e6,
    "device_GBps_compressed": raw.size / res["kernel_ms"] / 1e6, "crc_checked_on_device": not args.no_crc,
    "blocks_compared_with_zlib": checked,
    "host_reader_wall_s": t_host, "host_reader_GBps_inflated": inflated / t_host / 1e9,
    "host_inflate_stage_s": t_host_inflate, "host_inflate_stage_GBps": inflated / t_host_inflate / 1e9 if t_host_inflate else None,
    "host_cpus_effective": bench.effective_cpus(),
    "bam_gen_s": t_gen,
}
print(json.dumps(out))
bam.unlink()
tmp.rmdir()
