#!/usr/bin/env python3
"""tools/feed_soak.py -- GPU-box helper: BAMs whose RECORD STREAM is damaged (random bytes overwritten before the stream
is BGZF-compressed, so every block's CRC is right) through bin/pss-bam: the device-side record chain sees broken
block_size fields, cut-off and overlong records.  The command may fail (with a diagnosis) or fall back to the host
reader; it must not die of a signal or hang, and when it succeeds its tables must equal the host reader's.
    python3 tools/feed_soak.py [--files 60]"""
import argparse
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as ge  # noqa: E402
import pssbam_testlib as tl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--files", type=int, default=60)
args = ap.parse_args()
pkg = ge.load_pkg()
tmp = Path(tempfile.mkdtemp(prefix="pssbam_feed_soak_"))
contigs, refs, recs = tl.fuzz_dataset(123, 6000)
fa = tmp / "g.fa"
tl.write_fasta(fa, contigs)
good = tmp / "good.bam"
tl.write_bam_aligned(good, refs, recs, level=6)
data = bytearray(tl.bgzf_inflate(good.read_bytes()))
import struct
l_text = struct.unpack_from("<i", data, 4)[0]
n_ref = struct.unpack_from("<i", data, 8 + l_text)[0]
o = 12 + l_text
for _ in range(n_ref):
    ln = struct.unpack_from("<i", data, o)[0]
    o += 4 + ln + 4
header_end = o
rng = np.random.default_rng(77)
b = pkg.PKG_DIR / "bin" / "pss-bam"
outcomes = {"ok same tables": 0, "... of which stayed on the device feed": 0, "diagnosed failure": 0}
for f in range(args.files):
    d = bytearray(data)
    for _ in range(int(rng.integers(1, 6))):
        at = int(rng.integers(header_end, len(d) - 8))
        m = int(rng.integers(1, 6))
        d[at:at + m] = bytes(rng.integers(0, 256, m, dtype=np.uint8))
    blk = int(rng.choice([300, 5000, 0xFF00]))
    bam = tmp / "bad.bam"
    bam.write_bytes(b"".join(tl.bgzf_block(bytes(d[i:i + blk]), 6) for i in range(0, len(d), blk)) + tl.BGZF_EOF)
    res = []
    # device feed; host reader; device feed over two engines (alternating runs of the file: the record a run ends in is
    # handed from engine to engine, small super-batches so that it happens often)
    two = {"PSSBAM_NGPU": "2", "PSSBAM_OVERSUBSCRIBE": "1", "PSSBAM_CHUNK_BYTES": str(1 << 20), "PSSBAM_FEED_BATCH_BYTES": str(1 << 20),
           "PSSBAM_RUN_BATCHES": "1", "PSSBAM_FEED_SUPER_BYTES": str(1 << 20)}
    for tag, env in (("d", {}), ("h", {"PSSBAM_DEVICE_INFLATE": "0"}), ("t", two)):
        pr = subprocess.run([str(b), "-F", str(fa), "-B", str(bam), "-o", str(tmp / ("o" + tag)), "-r", "10"],
                            capture_output=True, text=True, env={**os.environ, "PSSBAM_STATS": "1", **env}, timeout=120)
        assert pr.returncode >= 0, (f, env, pr.returncode, pr.stderr[-400:])   # negative: killed by a signal
        res.append(pr)
    tab = lambda tag: (tmp / f"o{tag}.pss.counts.txt").read_text().split("\n", 6)[-1]
    if all(r.returncode == 0 for r in res):
        assert tab("d") == tab("h") == tab("t"), f
        outcomes["ok same tables"] += 1
        outcomes["... of which stayed on the device feed"] += int("host reader" not in res[0].stderr)
    else:
        assert all(r.returncode != 0 for r in res), (f, [r.returncode for r in res], [r.stderr[-300:] for r in res])
        outcomes["diagnosed failure"] += 1
print(outcomes, flush=True)
print("feed soak ok")
