#!/usr/bin/env python3
"""tools/timeline.py -- condenses a rocprofv3 kernel trace (+ memory-copy trace, if there is one) of ONE command
into a timeline: per bin of --bin ms, the share of the bin in which a kernel of each group was running and the bytes
copied host-to-device; then the busy time (union of all kernel intervals), the idle gaps above 2 ms, and whether the
inflate launches overlapped the genome kernels.  Used for profiles/r03_e2e_timeline.txt (tools/e2e_prof.sh).
    python3 tools/timeline.py <dir with *kernel_trace.csv> [--bin 20]"""
import argparse
import csv
import glob
import sys

ap = argparse.ArgumentParser()
ap.add_argument("root")
ap.add_argument("--bin", type=float, default=20.0)
args = ap.parse_args()

GROUPS = [("inflate", ("bgzf_inflate",)), ("crc", ("bgzf_crc",)), ("index", ("bgzf_chain", "bgzf_index")),
          ("tally", ("tally_", "reduce_partials")), ("genome", ("encode_genome", "pack_genome"))]


def group_of(name):
    for g, pats in GROUPS:
        if any(p in name for p in pats):
            return g
    return "other"


kt = sorted(glob.glob(f"{args.root}/**/*kernel_trace.csv", recursive=True))
if not kt:
    sys.exit(f"no *kernel_trace.csv under {args.root}")
ker = []
for f in kt:
    for row in csv.DictReader(open(f)):
        ker.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), group_of(row["Kernel_Name"]), row["Kernel_Name"]))
ker.sort()
cp = []
for f in glob.glob(f"{args.root}/**/*memory_copy_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        d = row.get("Direction", "")
        cp.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), d, int(row.get("Bytes", row.get("Size", 0)) or 0)))
t0 = min([k[0] for k in ker] + [c[0] for c in cp])
t1 = max([k[1] for k in ker] + [c[1] for c in cp])
binns = args.bin * 1e6
nb = int((t1 - t0) / binns) + 1
names = [g for g, _ in GROUPS] + ["other"]
occ = {g: [0.0] * nb for g in names}
h2d = [0.0] * nb


def spread(a, b, arr, weight=None):
    i = int((a - t0) / binns)
    while a < b and i < nb:
        e = min(b, t0 + (i + 1) * binns)
        arr[i] += (e - a) if weight is None else weight * (e - a) / max(b - (a if weight is None else a), 1)
        a = e
        i += 1


for a, b, g, _ in ker:
    spread(a, b, occ[g])
for a, b, d, n in cp:   # (the trace has no byte counts: the share of the bin with a host-to-device copy in flight, copies summed)
    if "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper():
        spread(a, b, h2d)
print(f"# {len(ker)} kernel dispatches, {len(cp)} copies; t = 0 at the first of them; bins of {args.bin:g} ms; "
      f"columns = share of the bin with a kernel of that group running / a host-to-device copy in flight (streams overlap: shares can add up past 1)")
print("t_ms   " + "".join(f"{g:>9s}" for g in names) + "   h2d_copy")
for i in range(nb):
    print(f"{i * args.bin:6.0f} " + "".join(f"{occ[g][i] / binns:9.2f}" for g in names) + f"   {h2d[i] / binns:8.2f}")
# union of kernel intervals
busy, cur_a, cur_b, gaps = 0, None, None, []
for a, b, g, n in ker:
    if cur_b is None or a > cur_b:
        if cur_b is not None:
            busy += cur_b - cur_a
            if a - cur_b > 2e6:
                gaps.append((cur_b - t0, a - cur_b, n))
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
span = ker[-1][1] - ker[0][0] if ker else 0
print(f"# kernels: first start {(ker[0][0] - t0) / 1e6:.1f} ms, last end {(max(k[1] for k in ker) - t0) / 1e6:.1f} ms; busy (union) {busy / 1e6:.1f} ms "
      f"of that {span / 1e6:.1f} ms span = {busy / max(span, 1):.2f}")
for g in names:
    tot = sum(b - a for a, b, gg, _ in ker if gg == g)
    cnt = sum(1 for k in ker if k[2] == g)
    if cnt:
        print(f"#   {g:8s} {cnt:5d} dispatches, {tot / 1e6:8.1f} ms in all")
for at, ln, nxt in gaps:
    print(f"# idle gap of {ln / 1e6:.1f} ms at t = {at / 1e6:.1f} ms (next: {nxt[:50]})")
gen = [(a, b) for a, b, g, _ in ker if g == "genome"]
inf = [(a, b) for a, b, g, _ in ker if g == "inflate"]
if gen and inf:
    g_a, g_b = min(a for a, _ in gen), max(b for _, b in gen)
    before = sum(1 for a, b in inf if a < g_b)
    over = sum(max(0, min(b, g_b) - max(a, g_a)) for a, b in inf)
    print(f"# genome kernels ran from {(g_a - t0) / 1e6:.1f} to {(g_b - t0) / 1e6:.1f} ms; {before} inflate launches had STARTED before they ended, "
          f"{over / 1e6:.1f} ms of inflate ran beside them")
if cp:
    tot = sum(b - a for a, b, d, n in cp if "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper())
    print(f"# host-to-device copies: {tot / 1e6:.1f} ms of copy time between {(min(c[0] for c in cp) - t0) / 1e6:.1f} and {(max(c[1] for c in cp) - t0) / 1e6:.1f} ms")
