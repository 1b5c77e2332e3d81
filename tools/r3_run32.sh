#!/bin/bash
set -u
OUT=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kside.py tests/test_gpu_sparse_parity.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
for i in 1 2; do timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3U_bench$i.json; done
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3U_c4.json
python - <<'PY'
import json
for f in ("bench1","bench2","c4"):
    d=json.load(open("gpurun_out/r3U_%s.json"%f))
    print(f, d["ms_per_step"], "outside", round(d["roofline"]["outside_events_ms_per_step"],3), d["roofline"]["pairs_per_sample"]["proof_round_int8"])
PY
