#!/usr/bin/env python3
"""tools/feed_scan.py -- GPU-box helper: bin/pss-bam on ONE generated BAM + FASTA of the benchmark shape under
several feed settings (environment variables), wall seconds each.
    python3 tools/feed_scan.py [--reads 200000000] [--scale-genome 1.0] -- "VAR=val VAR2=val" "VAR=val" ..."""
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
sys.path.insert(0, str(ROOT / "tests"))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

argv = sys.argv[1:]
settings = [""]
if "--" in argv:
    k = argv.index("--")
    settings = argv[k + 1:]
    argv = argv[:k]
ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=200_000_000)
ap.add_argument("--scale-genome", type=float, default=1.0)
ap.add_argument("--config", default="C3")
ap.add_argument("--level", type=int, default=1)
ap.add_argument("--ragged", action="store_true", help="BGZF blocks cut every 0xff00 bytes regardless of records (htsjdk-style)")
ap.add_argument("--quals", default="const", choices=["const", "binned", "full"], help="QUAL model of the generated BAM (synth.bam_file_host)")
args = ap.parse_args(argv)
pkg = ge.load_pkg()
from pss_bam_amd import synth  # noqa: E402

threads = bench.worker_threads()
d = synth.config(args.config, scale_genome=args.scale_genome)
region_len = d.pop("region_len")
d.pop("klen", None)
cfg = synth.make_cfg(**d)
tmp = Path(tempfile.mkdtemp(prefix="pssbam_scan_", dir=os.environ.get("TMPDIR", "/tmp")))
fa, bam = tmp / "ref.fa", tmp / "reads.bam"
synth.fasta_host(cfg, fa, threads=threads)
synth.bam_file_host(cfg, 0, args.reads, bam, level=args.level, threads=threads, ragged=args.ragged, quals=args.quals)
print(f"[feed_scan] {args.reads} reads, BAM {bam.stat().st_size / 1e9:.2f} GB, FASTA {fa.stat().st_size / 1e9:.2f} GB", flush=True)
ref_counts = None
for st in settings:
    env = {**os.environ, "PSSBAM_STATS": "1"}
    for kv in st.split():
        k, v = kv.split("=", 1)
        env[k] = v
    best = None
    for rep in range(int(os.environ.get("SCAN_REPS", "2"))):
        time.sleep(float(os.environ.get("SCAN_GAP_S", "1.0")))   # the previous run's worker / teardown is out of the way
        t = time.perf_counter()
        pr = subprocess.run([str(pkg.PKG_DIR / "bin" / "pss-bam"), "-F", str(fa), "-B", str(bam), "-o", str(tmp / "o"), "-r", str(region_len)],
                            capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t
        assert pr.returncode == 0, pr.stderr[-2000:]
        if best is None or wall < best[0]:
            best = (wall, pr.stderr)
    counts = "\n".join((tmp / "o.pss.counts.txt").read_text().splitlines()[6:])
    ref_counts = ref_counts or counts
    feed = re.search(r"device feed: (.*)\n", best[1])
    thr = re.search(r"device feed, this thread: (.*)\n", best[1])
    ph = re.search(r"phases: (.*)\n", best[1])
    ef = re.search(r"engine feed: (.*)\n", best[1])
    mn = re.search(r"main\(\) to reports written: ([\d.]+) s", best[1])
    fl = re.search(r"fasta load ([\d.]+) s", best[1])
    tally = re.search(r"total_s=([\d.]+)", best[1])
    print(json.dumps({"setting": st or "(default)", "wall_s": round(best[0], 3), "reads_per_s": round(args.reads / best[0]),
                      "tally_phase_s": float(tally.group(1)) if tally else None, "same_tables": counts == ref_counts,
                      "device_feed": feed.group(1) if feed else None, "feed_thread": thr.group(1) if thr else None,
                      "phases": ph.group(1) if ph else None,
                      "engine_feed": ef.group(1) if ef else None, "main_to_reports_s": float(mn.group(1)) if mn else None,
                      "fasta_load_s": float(fl.group(1)) if fl else None,
                      "early_feed": (lambda m_: m_.group(1) if m_ else None)(re.search(r"early feed \(helper thread[^:]*\): (.*)\n", best[1])),
                      "process": (lambda m_: m_.group(1) if m_ else None)(re.search(r"process creation to main\(\): (.*)\n", best[1])),
                      "device": (lambda m_: m_.group(1) if m_ else None)(re.search(r"\[pssbam\] device: (.*)\n", best[1])),
                      "engine_create": re.findall(r"engine on device (.*)\n", best[1])}), flush=True)
for p in tmp.iterdir():
    p.unlink()
tmp.rmdir()
