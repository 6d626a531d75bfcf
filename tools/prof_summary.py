#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + counter passes) for the tally kernels."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(f"{root}/stats/**/*kernel_stats.csv", recursive=True)):
    print("==", f)
    for row in list(csv.DictReader(open(f)))[:8]:
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage")})
for d in sorted(glob.glob(f"{root}/pmc*")):
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[(k, row["Counter_Name"])] += 1
        print("==", f)
        for k, v in acc.items():
            if "tally" in k or "encode" in k:
                print(" ", k)
                for c, val in v.items():
                    print(f"    {c:28s} sum={val:.4g}  per-dispatch={val / cnt[(k, c)]:.4g}  (n={cnt[(k, c)]})")
