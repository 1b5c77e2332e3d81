#!/bin/bash
# round 3, second GPU call: proof for all candidates + margin 0, collective policy tests, full-size sparse tests, trace
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_proof.py tests/test_gpu_sharded.py -x -q -m gpu > $OUT/r3b_tests1.log 2>&1; tail -12 $OUT/r3b_tests1.log
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3b_bench.json 2> $OUT/r3b_bench.err; tail -c 400 $OUT/r3b_bench.err
GMMVB_SETTLE_MARGIN=5 timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3b_bench_m5.json 2> $OUT/r3b_bench_m5.err; tail -c 400 $OUT/r3b_bench_m5.err
python - <<'PY'
import json
for n in ("","_m5"):
    try:
        d=json.load(open(f"gpurun_out/r3b_bench{n}.json"))
        print(n, round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
        for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
        p=d["per_step"]
        for k in ("wall_ms","estep_ms","mstep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
        print("   kern", [k[6:12] for k in p["estep_kernel"]])
    except Exception as e: print(n, "failed", e)
PY
rm -rf $OUT/r3b_trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3b_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/r3b_bench_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/r3b_trace.err)
python tools/summarize_rocprof.py $OUT/r3b_trace > $OUT/r3b_kernel_summary.md 2>> $OUT/r3b_trace.err; head -50 $OUT/r3b_kernel_summary.md
find $OUT/r3b_trace -name "*kernel_trace.csv" -size +20M -delete
timeout 1500 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu > $OUT/r3b_tests_full.log 2>&1; tail -15 $OUT/r3b_tests_full.log
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py -x -q -m gpu > $OUT/r3b_tests2.log 2>&1; tail -8 $OUT/r3b_tests2.log
