#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3o_bench.json 2> $OUT/r3o_bench.err; tail -c 200 $OUT/r3o_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3o_bench.json"))
print(d["ms_per_step"], d["per_step"]["wall_ms"]); print(d["per_step"]["sweep_share_of_bound_array"]); print(d["per_step"]["proof_pairs_per_sample"])
PY
