#!/usr/bin/env python3
"""Per-call durations of the kernels whose names contain one of the given substrings, in dispatch order:
    python tools/trace_calls.py <rocprofv3 output dir> <substring> [<substring> ...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
t0 = None
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if any(k in n for k in sys.argv[2:]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        t0 = s if t0 is None else t0
        print(round((s - t0) / 1e6, 2), n.replace("void gmmvb::", "")[:60], round((e - s) / 1e6, 3))
