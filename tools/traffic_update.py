#!/usr/bin/env python3
"""tools/traffic_update.py -- merges the traffic entries of gpurun_out/prof_<TAG>/summary.json (tools/pmc_profile.py) into
profiles/traffic.json: an entry replaces the one of the same (config, order, k, N, launch size within 10 %), and is stamped
with the git SHA given on the command line.  Also copies each summary to profiles/<prefix>_pmc_<TAG>.json and its kernel
stats to profiles/<prefix>_kernel_stats_<TAG>.csv.
    python3 tools/traffic_update.py r03 <git sha> TAG:config[:unsorted][:k=K] ..."""
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
prefix, sha = sys.argv[1], sys.argv[2]
tj = ROOT / "profiles" / "traffic.json"
doc = json.loads(tj.read_text())
REGION = {"C1": 15, "C2": 25, "C3": 25, "C4": 15, "C5": 25}
for spec in sys.argv[3:]:
    parts = spec.split(":")
    tag, config = parts[0], parts[1]
    unsorted_ = "unsorted" in parts[2:]
    klen = next((int(p[2:]) for p in parts[2:] if p.startswith("k=")), None)
    sm = ROOT / "gpurun_out" / f"prof_{tag}" / "summary.json"
    if not sm.exists():
        print(f"{tag}: no summary")
        continue
    summ = json.loads(sm.read_text())
    te = summ.get("traffic_entry")
    if not te:
        print(f"{tag}: no traffic entry")
        continue
    en = {"config": config, "unsorted": unsorted_, "klen": klen, "region_len": REGION[config], "git_sha": sha, **te}
    keep = []
    for old in doc["entries"]:
        same = (old["config"] == config and bool(old.get("unsorted", False)) == unsorted_ and old.get("klen") == klen
                and old.get("region_len") == REGION[config]
                and abs(old["reads_per_launch"] - en["reads_per_launch"]) <= 0.10 * en["reads_per_launch"])
        if not same:
            keep.append(old)
    keep.append(en)
    doc["entries"] = keep
    shutil.copy(sm, ROOT / "profiles" / f"{prefix}_pmc_{tag}.json")
    ks = sm.parent / "kernel_stats.csv"
    if ks.exists():
        shutil.copy(ks, ROOT / "profiles" / f"{prefix}_kernel_stats_{tag}.csv")
    print(f"{tag}: {en['hbm_bytes_per_read']:.1f} B/read at {en['reads_per_launch'] / 1e6:.2f} M reads per launch")
tj.write_text(json.dumps(doc, indent=1) + "\n")
