#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
for g in 1 2 3 4; do
  GMMVB_PROOF_GRID=$g timeout 300 python tools/bench_proof.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('grid x$g', d['ns_per_pair'], d['proof_ms_per_launch'])"
done
for g in 1 2 3; do
  GMMVB_PROOF_GRID=$g timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3u_bench_g$g.json 2>/dev/null
  GMMVB_PROOF_GRID=$g timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3u_c4_g$g.json 2>/dev/null
done
python - <<'PY'
import json
for g in (1,2,3):
    for f in ("bench","c4"):
        d=json.load(open("gpurun_out/r3u_%s_g%d.json"%(f,g)))
        print(f, g, d["ms_per_step"], d["roofline"]["kernel_groups"]["estep_proof"]["ms"], d["roofline"]["kernel_groups"]["estep_select"]["ms"])
PY
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_proof.py -x -q -m gpu 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
