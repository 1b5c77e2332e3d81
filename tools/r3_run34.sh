#!/bin/bash
set -u
OUT=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_sparse.py tests/test_gpu_full_size.py tests/test_gpu_sharded.py tests/test_gpu_tiled.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3Y_bench.json
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3Y_c4.json
timeout 900 python bench.py --config c4 --no-cpu --no-legs 2>/dev/null | grep -a "^{" > $OUT/r3Y_c4def.json
timeout 600 python bench.py --no-cpu --no-legs --steps 80 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3Y_w5s80.json
python - <<'PY'
import json
for f in ("bench","c4","c4def","w5s80"):
    d=json.load(open("gpurun_out/r3Y_%s.json"%f))
    print(f, d["ms_per_step"], d["per_step"]["wall_ms"][:6], d["per_step"]["wall_ms"][-3:], set(d["per_step"]["estep_kernel"]))
PY
