#!/usr/bin/env python3
"""Per-iteration kernel breakdown of a rocprofv3 kernel trace of bench.py: dispatches are split into VB iterations at every
kside_step_kernel launch; prints, for the chosen iterations, each kernel's summed time and the gaps between dispatches."""
import csv
import re
import sys

path = sys.argv[1]
want = [int(a) for a in sys.argv[2:]] or None
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        m = re.search(r"gmmvb::(\w+)", name)
        short = m.group(1) if m else re.sub(r"^void ", "", name)[:38]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
steps, cur = [], []
for s, e, n in rows:
    cur.append((s, e, n))
    if n.startswith("kside_step_kernel"):
        steps.append(cur)
        cur = []
for i, st in enumerate(steps):
    if want is not None and i not in want:
        continue
    span = (st[-1][1] - st[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in st) / 1e6
    agg = {}
    for s, e, n in st:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    print(f"--- iteration {i}: {len(st)} dispatches, span {span:.2f} ms, busy {busy:.2f} ms, idle {span - busy:.2f} ms")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t >= 0.02:
            print(f"   {n:40s} x{c:<3d} {t:7.3f} ms")
