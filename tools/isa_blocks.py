#!/usr/bin/env python3
"""tools/isa_blocks.py -- per-basic-block instruction mix of one kernel in a hipcc -S listing.
usage: isa_blocks.py engine.s <mangled-kernel-name-substring> [min_instrs]"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 15
start = next(i for i, l in enumerate(text) if l.startswith("_Z") and want in l and l.rstrip().endswith(("E:", ")")) or (l.startswith("_Z") and want in l and ":" in l))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
blocks, cur = [], None
for i in range(start, end + 1):
    l = text[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = dict(name=m.group(1) if m else "entry", line=i - start, valu=0, salu=0, lds=0, vmem=0, smem=0, br=[])
        blocks.append(cur)
        if m:
            continue
    t = l.strip().split(" ")[0] if l.strip() else ""
    if not t or t[0] in ";.":
        continue
    if t.startswith("v_"):
        cur["valu"] += 1
    elif t.startswith("ds_"):
        cur["lds"] += 1
    elif t.startswith(("global_", "buffer_", "scratch_", "flat_")):
        cur["vmem"] += 1
    elif t.startswith(("s_load", "s_buffer")):
        cur["smem"] += 1
    elif t.startswith("s_"):
        cur["salu"] += 1
        if "branch" in t:
            cur["br"].append(l.strip().split()[-1])
tot = dict(valu=0, salu=0, lds=0, vmem=0, smem=0)
for b in blocks:
    for k in tot:
        tot[k] += b[k]
    if b["valu"] + b["salu"] + b["lds"] + b["vmem"] + b["smem"] >= min_n:
        print(f'{b["name"]:>10} @{b["line"]:5d} valu {b["valu"]:4d} salu {b["salu"]:3d} lds {b["lds"]:3d} vmem {b["vmem"]:2d} smem {b["smem"]:2d} -> {",".join(b["br"][-2:])}')
print("total", tot)
