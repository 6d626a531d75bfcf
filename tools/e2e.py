#!/usr/bin/env python3
"""tools/e2e.py -- end-to-end (file -> tables) timing of the C front end on the GPU box:
a generated BGZF BAM + FASTA on local disk through bin/pss-bam, i.e. including FASTA load,
BGZF inflate on the host threads, PCIe H2D of every record byte, the kernels and the report.
Prints one JSON object.  (bench.py measures the HBM-resident hot path; this is the number
that includes the host feed, quoted in DESIGN.md and never used as bench `value`.)"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=10_000_000)
ap.add_argument("--scale-genome", type=float, default=0.1)
ap.add_argument("--level", type=int, default=1)
ap.add_argument("--config", default="C2")
ap.add_argument("--ragged", action="store_true", help="BGZF blocks cut every 0xff00 bytes regardless of records (default: htslib's layout)")
args = ap.parse_args()

pkg = ge.load_pkg()
from pss_bam_amd import synth  # noqa: E402

threads = os.cpu_count() or 8
d = synth.config(args.config, n_reads=args.reads, scale_genome=args.scale_genome)
region_len = d.pop("region_len")
d.pop("klen", None)
cfg = synth.make_cfg(**d)
tmp = Path(tempfile.mkdtemp(prefix="pssbam_e2e_", dir=os.environ.get("TMPDIR", "/tmp")))
fa, bam = tmp / "ref.fa", tmp / "reads.bam"
t = time.time()
synth.fasta_host(cfg, fa, threads=threads)
t_fa = time.time() - t
t = time.time()
synth.bam_file_host(cfg, 0, args.reads, bam, level=args.level, threads=threads, ragged=args.ragged)
t_bam = time.time() - t
env = {**os.environ, "PSSBAM_STATS": "1"}
t = time.time()
pr = subprocess.run([str(pkg.PKG_DIR / "bin" / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp / "out"), "-r",
                     str(region_len)], capture_output=True, text=True, env=env)
wall = time.time() - t
if pr.returncode != 0:
    print(pr.stderr[-2000:])
    raise SystemExit(1)
m = re.search(r"gpus=(\d+) inflate_s=([\d.]+) total_s=([\d.]+)", pr.stderr)
inflate_s, tally_s = float(m.group(2)), float(m.group(3))
phases = re.search(r"phases: (.*) s\n", pr.stderr)
reader = re.search(r"reader thread: (.*) s\n", pr.stderr)
teardown = re.search(r"teardown ([\d.]+) s", pr.stderr)
rec_bytes = sum(int(x) for x in synth.sizes_host(cfg, 0, 1000, threads=1)) / 1000 * args.reads
print(json.dumps({
    "reads": args.reads, "bam_bytes": bam.stat().st_size, "inflated_record_bytes": rec_bytes, "fasta_bytes": fa.stat().st_size,
    "deflate_level": args.level, "block_layout": "ragged" if args.ragged else "htslib", "host_threads": threads,
    "wall_s_whole_command": wall, "tally_phase_s": tally_s, "inflate_s": inflate_s,
    "genome_load_and_upload_s": wall - tally_s,
    "reads_per_s_tally_phase": args.reads / tally_s, "reads_per_s_whole_command": args.reads / wall,
    "inflate_GBps": rec_bytes / inflate_s / 1e9 if inflate_s else None,
    "workload_gen_s": {"fasta": t_fa, "bam": t_bam}, "phases": phases.group(1) if phases else None,
    "reader_thread": reader.group(1) if reader else None, "teardown_s": float(teardown.group(1)) if teardown else None,
}))
for p in tmp.iterdir():
    p.unlink()
tmp.rmdir()
