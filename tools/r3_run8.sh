#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_generic.py tests/test_mvn.py tests/test_gpu_edges.py -x -q -m gpu > $OUT/r3h_tests1.log 2>&1; tail -15 $OUT/r3h_tests1.log
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_sparse.py tests/test_gpu_parity.py -x -q -m gpu > $OUT/r3h_tests2.log 2>&1; tail -8 $OUT/r3h_tests2.log
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3h_bench.json 2> $OUT/r3h_bench.err; tail -c 300 $OUT/r3h_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3h_bench.json"))
print(round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
p=d["per_step"]
for k in ("wall_ms","estep_ms","mstep_ms"): print("  ",k,p[k])
PY
