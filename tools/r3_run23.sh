#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_sparse.py tests/test_gpu_proof.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3x_bench.json 2> $OUT/r3x_bench.err; tail -c 200 $OUT/r3x_bench.err
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3x_c4.json 2> $OUT/r3x_c4.err
python - <<'PY'
import json
for f in ("bench","c4"):
    d=json.load(open("gpurun_out/r3x_%s.json"%f))
    g=d["roofline"]["kernel_groups"]
    print(f, d["ms_per_step"], "select", g["estep_select"]["ms"], "lse_mask", g["estep_lse_mask"]["ms"], d["per_step"]["wall_ms"][-6:], d["roofline"]["pairs_per_sample"])
PY
