#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats --output-format csv` run for profiles/.

usage: summarize_rocprof.py <dir with *_kernel_trace.csv> [--pmc <dir with *_counter_collection.csv>]

Groups gmmvb kernel dispatches by (kernel, grid size) so that the timed-region launches of bench.py
(the large grids) are not averaged with the small parity-gate launches, and prints everything else
(torch / rocBLAS / rocSOLVER K-side kernels) as one aggregate line per kernel.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    if name.startswith("gmmvb::"):
        return name.split("(")[0]
    return name.split("(")[0][:70]


def main():
    d = sys.argv[1]
    trace = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    groups = defaultdict(list)
    meta = {}
    total = 0.0
    with open(trace) as f:
        for row in csv.DictReader(f):
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
            total += dur
            name = short(row["Kernel_Name"])
            wgs = int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"]))
            # same grid can serve very different row counts (the M-step's split count saturates): bucket by decade
            key = (name, wgs, len(str(int(dur * 1000)))) if name.startswith("gmmvb::") else (name, 0, 0)
            groups[key].append(dur)
            meta[key] = (row["VGPR_Count"], row["Accum_VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Scratch_Size"])
    print(f"# source: {os.path.relpath(trace)}")
    print(f"# total kernel time {total:.1f} ms over {sum(len(v) for v in groups.values())} dispatches\n")
    print("| kernel | workgroups | calls | avg ms | min ms | max ms | total ms | % | VGPR | AGPR | SGPR | LDS B | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / total < 0.0005 and not key[0].startswith("gmmvb::"):
            continue
        m = meta[key]
        print(f"| {key[0]} | {key[1] or ''} | {len(v)} | {sum(v)/len(v):.3f} | {min(v):.3f} | {max(v):.3f} | {sum(v):.1f} | "
              f"{100*sum(v)/total:.2f} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} |")


if __name__ == "__main__":
    main()
