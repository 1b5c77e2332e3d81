#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_sparse.py tests/test_gpu_full_size.py -x -q -m gpu > $OUT/r3j_tests1.log 2>&1; tail -8 $OUT/r3j_tests1.log
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3j_bench.json 2> $OUT/r3j_bench.err; tail -c 300 $OUT/r3j_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3j_bench.json"))
print(round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
p=d["per_step"]
for k in ("wall_ms","estep_ms","mstep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
PY
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3j_bench_c4.json 2> $OUT/r3j_c4.err; tail -c 300 $OUT/r3j_c4.err
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3j_bench_c4.json"))
    print("c4", round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
    p=d["per_step"]
    for k in ("wall_ms","estep_ms","proof_pairs_per_sample"): print("  ",k,p[k])
except Exception as e: print("c4 failed", e)
PY
