#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3t_bench.json 2> $OUT/r3t_bench.err; tail -c 200 $OUT/r3t_bench.err
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3t_c4.json 2> $OUT/r3t_c4.err
python - <<'PY'
import json
for f in ("bench","c4"):
    d=json.load(open("gpurun_out/r3t_%s.json"%f))
    print(f, d["ms_per_step"], d["per_step"]["wall_ms"]); print(d["per_step"]["estep_ms"]); print(d["per_step"].get("sweep_share_of_bound_array")); print(d["per_step"]["proof_pairs_per_sample"]); print(d["roofline"]["pairs_per_sample"])
PY
