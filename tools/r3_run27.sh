#!/bin/bash
set -u
OUT=gpurun_out
timeout 300 python tools/bench_proof.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('proof ns/pair', d['ns_per_pair'], d['proof_ms_per_launch'])"
timeout 900 python -m pytest tests/test_gpu_proof.py tests/test_gpu_sparse.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3H_bench.json
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3H_c4.json
timeout 600 python bench.py --no-cpu --no-legs 2>/dev/null | grep -a "^{" > $OUT/r3H_bench_def.json
python - <<'PY'
import json
for f in ("bench","c4","bench_def"):
    d=json.load(open("gpurun_out/r3H_%s.json"%f))
    g=d["roofline"]["kernel_groups"]
    print(f, d["ms_per_step"], {k:round(v["ms"],2) for k,v in g.items()}, d["per_step"]["estep_ms"][:3])
PY
