#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_sparse.py tests/test_gpu_sharded.py -x -q -m gpu > $OUT/r3f_tests1.log 2>&1; tail -12 $OUT/r3f_tests1.log
for cfg in "def:" "m0:GMMVB_SETTLE_MARGIN=0"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3f_bench_$tag.json 2> $OUT/r3f_bench_$tag.err; tail -c 300 $OUT/r3f_bench_$tag.err
done
python - <<'PY'
import json
for n in ("def","m0"):
    try:
        d=json.load(open(f"gpurun_out/r3f_bench_{n}.json"))
        print(n, round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
        for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
        p=d["per_step"]
        for k in ("wall_ms","estep_ms","mstep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
    except Exception as e: print(n, "failed", e)
PY
rm -rf $OUT/r3f_pmc; (cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "estep_i8_proof" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3f_pmc -- python3 $GRAFT_REPO_ROOT/tools/bench_proof.py > $GRAFT_REPO_ROOT/$OUT/r3f_pmc.json 2> $GRAFT_REPO_ROOT/$OUT/r3f_pmc.err); tail -c 600 $OUT/r3f_pmc.err; ls $OUT/r3f_pmc/*/ 2>/dev/null | head; head -c 3000 $OUT/r3f_pmc/*/*counter_collection.csv 2>/dev/null
