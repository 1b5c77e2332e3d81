#!/bin/bash
# lazy sweep: per-iteration kernel times and the sweep's traffic
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/r3n_trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3n_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/r3n_bench_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/r3n_trace.err)
python tools/summarize_rocprof.py $OUT/r3n_trace > $OUT/r3n_kernel_summary.md 2>> $OUT/r3n_trace.err
python tools/trace_steps.py $OUT/r3n_trace/runc/*_kernel_trace.csv 6 9 12 16 24 > $OUT/r3n_steps.txt 2>&1; grep -v "scan_\|pack_\|copyBuffer\|kside_finish\|mstep_plan\|at::native\|gather_plan\|sum_parts" $OUT/r3n_steps.txt | head -120
python - <<'PY'
import csv, glob, re
rows=[]
for f in glob.glob("gpurun_out/r3n_trace/runc/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"gmmvb::(\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else "other"))
rows.sort()
print("sweep ms per launch:", [round((e-s)/1e6,3) for s,e,n in rows if n.startswith("rec_sweep")])
PY
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $OUT/r3n_pmc_$c
  (cd /tmp && timeout 900 rocprofv3 --pmc $c --kernel-include-regex "rec_sweep|rec_finish|rec_proof_decide|fill_lists|settled_mask" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3n_pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/r3n_pmc_$c.err); done
python tools/summarize_pmc.py $OUT/r3n_pmc_FETCH_SIZE $OUT/r3n_pmc_WRITE_SIZE > $OUT/r3n_pmc_summary.md 2> $OUT/r3n_pmc.err; head -40 $OUT/r3n_pmc_summary.md
find $OUT/r3n_pmc_FETCH_SIZE $OUT/r3n_pmc_WRITE_SIZE $OUT/r3n_trace -name "*.csv" -size +20M -delete
