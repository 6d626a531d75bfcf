#!/usr/bin/env python3
"""tools/pmc_profile.py -- GPU-box helper: rocprofv3 evidence for ONE bench configuration.

    python3 tools/pmc_profile.py TAG -- <bench.py arguments>

Runs `python3 bench.py <args> --no-cpu-baseline --no-e2e` under
  1. rocprofv3 --kernel-trace --stats                       (per-kernel durations)
  2. rocprofv3 --pmc FETCH_SIZE                             (own pass: FETCH_SIZE takes 3 of the 4 TCC slots)
  3. rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  4. rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
  5. rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
each counter group in its own run, never together with a trace domain other than --kernel-trace
(MI355X_MICROARCH.md, "HBM" / "rocprofv3 PMC slots").  Writes gpurun_out/prof_TAG/summary.json:
per-launch figures of the dominant tally kernel, the bench line of every pass, and a ready-made
entry for profiles/traffic.json:
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes      (gfx950: FETCH_SIZE reports half of
                a 16 B/lane streaming read; the kernel's traffic is dominated by the LDS-DMA
                record stream, so the guide's doubling applies; TCC_MISS * 128 B is printed
                beside it as the cross-check)."""
import csv
import glob
import json
import os
import subprocess
import sys
import time
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import kernel_blobs  # noqa: E402
tag = sys.argv[1]
assert sys.argv[2] == "--"
bench_args = sys.argv[3:] + ["--no-cpu-baseline", "--no-e2e"]
out = ROOT / "gpurun_out" / f"prof_{tag}"
out.mkdir(parents=True, exist_ok=True)
os.chdir(ROOT)
env = {**os.environ, "TMPDIR": "/tmp"}
cmd = ["python3", "bench.py"] + bench_args

PASSES = [
    ("stats", ["--kernel-trace", "--stats"]),
    ("fetch", ["--kernel-trace", "--pmc", "FETCH_SIZE"]),
    ("write", ["--kernel-trace", "--pmc", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]),
    ("sq1", ["--kernel-trace", "--pmc", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
             "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"]),
    ("sq2", ["--kernel-trace", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VMEM"]),
]
only = os.environ.get("PMC_PASSES")
summary = {"tag": tag, "command": " ".join(cmd), "date": time.strftime("%Y-%m-%d"), "passes": {}}
for name, flags in PASSES:
    if only and name not in only.split(","):
        continue
    d = out / name
    t0 = time.time()
    pr = subprocess.run(["rocprofv3"] + flags + ["--output-format", "csv", "-d", str(d), "--"] + cmd, env=env,
                        capture_output=True, text=True)
    print(f"[pmc_profile] pass {name}: rc {pr.returncode}, {time.time() - t0:.0f}s", flush=True)
    line = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    rec = {"rc": pr.returncode}
    if line:
        b = json.loads(line[-1])
        rec["bench"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "roofline": b["roofline"],
                        "launches_per_step": b["config"]["launches_per_step"], "reads_rank0": b["config"]["reads_rank0"]}
        summary["reads_per_launch"] = b["config"]["reads_rank0"] / b["config"]["launches_per_step"]
    else:
        rec["stderr_tail"] = pr.stderr[-1500:]
    if name == "stats":
        for f in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            rec["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage") if k in r}
                                   for r in rows[:8]]
            (out / "kernel_stats.csv").write_text(open(f).read())
    else:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            acc, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[(k, row["Counter_Name"])] += 1
            rec["counters"] = {}
            for k, v in acc.items():
                if "tally_tiled" in k or "tally_simple" in k or "tally_compact" in k:
                    rec["counters"][k[:90]] = {c: {"per_dispatch": val / cnt[(k, c)], "dispatches": cnt[(k, c)]}
                                               for c, val in v.items()}
    summary["passes"][name] = rec


def main_kernel(passname):
    cs = summary["passes"].get(passname, {}).get("counters", {})
    best = None
    for k, v in cs.items():   # the variant with the most dispatches is the timed one
        n = max(x["dispatches"] for x in v.values())
        if best is None or n > best[0]:
            best = (n, k, v)
    return best


f, w = main_kernel("fetch"), main_kernel("write")
if f and w and "reads_per_launch" in summary:
    fetch_kb = f[2]["FETCH_SIZE"]["per_dispatch"]
    write_kb = w[2]["WRITE_SIZE"]["per_dispatch"]
    miss = w[2].get("TCC_MISS_sum", {}).get("per_dispatch")
    hit = w[2].get("TCC_HIT_sum", {}).get("per_dispatch")
    rpl = summary["reads_per_launch"]
    hbm = (2.0 * fetch_kb + write_kb) * 1024.0
    summary["traffic_entry"] = {
        "reads_per_launch": rpl, "hbm_bytes_per_read": hbm / rpl, "fetch_size_kb_per_launch": fetch_kb,
        "write_size_kb_per_launch": write_kb, "tcc_miss_x128_bytes_per_read": miss * 128.0 / rpl if miss else None,
        "l2_hit_rate": hit / (hit + miss) if hit is not None and miss else None,
        "kernel": f[1], "command": " ".join(cmd), "date": summary["date"],
        "kernel_blobs": kernel_blobs(),   # git blob ids of the kernel sources measured (bench.py: roofline.traffic_stale)
    }
(out / "summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps({k: summary.get(k) for k in ("tag", "reads_per_launch", "traffic_entry")}, indent=1))
st = summary["passes"].get("stats", {}).get("kernel_stats")
if st:
    for r in st[:4]:
        print(r)
for p in ("sq1", "sq2"):
    mk = main_kernel(p)
    if mk:
        print(p, mk[1][:60], {c: round(x["per_dispatch"]) for c, x in mk[2].items()})
