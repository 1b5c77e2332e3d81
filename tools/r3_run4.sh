#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_sharded.py tests/test_gpu_tiled.py tests/test_gpu_sparse.py -x -q -m gpu > $OUT/r3d_tests1.log 2>&1; tail -12 $OUT/r3d_tests1.log
for cfg in "def:" "tb2:GMMVB_X_TB=2" "rg4:GMMVB_X_REGROUP_ACT=4" "tb2rg4:GMMVB_X_TB=2 GMMVB_X_REGROUP_ACT=4"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3d_bench_$tag.json 2> $OUT/r3d_bench_$tag.err; tail -c 300 $OUT/r3d_bench_$tag.err
done
python - <<'PY'
import json
for n in ("def","tb2","rg4","tb2rg4"):
    try:
        d=json.load(open(f"gpurun_out/r3d_bench_{n}.json"))
        print(n, round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
        for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
        p=d["per_step"]
        for k in ("wall_ms","estep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
        print("   kern", [k[6:12] for k in p["estep_kernel"]])
    except Exception as e: print(n, "failed", e)
PY
rm -rf $OUT/r3d_trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3d_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/r3d_bench_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/r3d_trace.err)
python tools/summarize_rocprof.py $OUT/r3d_trace > $OUT/r3d_kernel_summary.md 2>> $OUT/r3d_trace.err
python tools/trace_steps.py $OUT/r3d_trace/runc/*_kernel_trace.csv 2 3 4 5 6 7 8 9 10 > $OUT/r3d_steps.txt 2>&1; grep -v "scan_\|pack_\|copyBuffer\|kside_finish\|mstep_plan\|at::native" $OUT/r3d_steps.txt | head -150
timeout 900 python tools/full_run.py > $OUT/r3d_full_run.json 2> $OUT/r3d_full.err; tail -c 300 $OUT/r3d_full.err; head -c 900 $OUT/r3d_full_run.json; echo
timeout 900 python bench.py --config c4 --scaling strong --gpus 1 --no-cpu --no-legs --steps 3 --warmup 2 > $OUT/r3d_bench_c4_strong1.json 2> $OUT/r3d_c4s.err; tail -c 600 $OUT/r3d_c4s.err; head -c 1200 $OUT/r3d_bench_c4_strong1.json; echo
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_proof.py -x -q -m gpu > $OUT/r3d_tests2.log 2>&1; tail -8 $OUT/r3d_tests2.log
