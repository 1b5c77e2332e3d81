#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/r3k_trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3k_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/r3k_bench_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/r3k_trace.err)
python tools/summarize_rocprof.py $OUT/r3k_trace > $OUT/r3k_kernel_summary.md 2>> $OUT/r3k_trace.err
python tools/trace_steps.py $OUT/r3k_trace/runc/*_kernel_trace.csv 6 7 9 12 16 24 > $OUT/r3k_steps.txt 2>&1; grep -v "scan_\|pack_\|copyBuffer\|kside_finish\|mstep_plan\|at::native\|gather_plan\|sum_parts" $OUT/r3k_steps.txt | head -120
python - <<'PY'
import csv, glob, re
# per-dispatch list of the two mstep_list launches in iterations 6..9
rows=[]
for f in glob.glob("gpurun_out/r3k_trace/runc/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"gmmvb::(\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else "other", r["Grid_Size_X"]))
rows.sort()
it=0
for s,e,n,g in rows:
    if n.startswith("kside_step"): it+=1
    if 6<=it<=9 and n.startswith(("mstep_list","reduce_chunks","estep_gather","estep_i8_proof")):
        print(it, n, g, round((e-s)/1e6,3))
PY
