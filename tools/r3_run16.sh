#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3p_c4_lazy.json 2> $OUT/r3p_c4_lazy.err; tail -c 200 $OUT/r3p_c4_lazy.err
GMMVB_SWEEP_LAZY=0 timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3p_c4_nolazy.json 2> $OUT/r3p_c4_nolazy.err
python - <<'PY'
import json
for f in ("lazy","nolazy"):
    d=json.load(open("gpurun_out/r3p_c4_%s.json"%f))
    print(f, d["ms_per_step"], d["per_step"]["wall_ms"]); print(d["per_step"]["estep_ms"]); print(d["per_step"].get("sweep_share_of_bound_array")); print(d["per_step"]["proof_pairs_per_sample"]); print(d["per_step"]["estep_kernel"])
PY
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -5
