#!/bin/bash
set -u
OUT=gpurun_out
bash tools/trace_c4.sh r3r
python tools/trace_steps.py $OUT/r3r_c4_trace/runc/*_kernel_trace.csv 7 10 16 24 > $OUT/r3r_c4_steps.txt 2>&1; grep -v "scan_\|pack_\|copyBuffer\|kside_finish\|mstep_plan\|at::native\|gather_plan\|sum_parts" $OUT/r3r_c4_steps.txt | head -70
timeout 600 python tools/bench_hmm.py --classes 128 --degree 8 --rows 200000 --steps 3 --warmup 1 --no-cpu 2>&1 | tail -2 | cut -c1-600
