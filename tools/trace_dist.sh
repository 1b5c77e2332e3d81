#!/bin/bash
# kernel trace of the bench with --force-dist (RCCL group of one rank): gpurun_out/<tag>_dist_trace
TAG=$1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_dist_trace
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_dist_trace -- python3 $GRAFT_REPO_ROOT/bench.py --force-dist --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/${TAG}_dist_line.json 2> $OUT/${TAG}_dist_trace.err)
python3 - <<P
import csv, glob
f = glob.glob("$OUT/${TAG}_dist_trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rec_sweep" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev = None
for r in rows[a:b + 1]:
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0
    if gap > 20 or (e - s) > 200e3:
        print(f"{(s - t0) / 1e6:8.3f} ms  dur {(e - s) / 1e3:8.1f} us gap {gap:7.1f} us  {r['Kernel_Name'][:60]}")
    prev = e
P
