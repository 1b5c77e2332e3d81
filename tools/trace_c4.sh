#!/bin/bash
# kernel trace of bench.py --config c4 (K=256, D=64, 1.25e7 rows)
TAG=$1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_c4_trace
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_c4_trace -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/${TAG}_c4_line_profiled.json 2> $OUT/${TAG}_c4_trace.err)
python3 tools/summarize_rocprof.py $OUT/${TAG}_c4_trace > $OUT/${TAG}_c4_kernel_summary.md 2>> $OUT/${TAG}_c4_trace.err
head -26 $OUT/${TAG}_c4_kernel_summary.md
python3 -c "
import json; d=json.load(open('$OUT/${TAG}_c4_line_profiled.json')); p=d['per_step']; print(d['ms_per_step']); print(p['estep_ms']); print(p['mstep_ms']); print(p['evaluated_components_per_sample']); print(p['active_components_per_sample']); print(p['estep_kernel'])"
find $OUT/${TAG}_c4_trace -name "*kernel_trace.csv" -size +20M -delete
