#!/bin/bash
set -u
OUT=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kside.py tests/test_gpu_parity.py tests/test_gpu_sparse_parity.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
timeout 600 python bench.py --config c2 --no-cpu --no-legs 2>/dev/null | grep -a "^{" > $OUT/r3M_c2.json
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3M_c4.json
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3M_bench.json
timeout 300 python tools/bench_small.py 2>/dev/null | tail -1 | cut -c1-400
python - <<'PY'
import json
for f in ("c2","c4","bench"):
    d=json.load(open("gpurun_out/r3M_%s.json"%f))
    print(f, d["ms_per_step"], d["roofline"]["outside_events_ms_per_step"])
PY
