#!/bin/bash
set -u
OUT=gpurun_out
for v in 0 1; do
  GMMVB_TB_FULL=$v timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3X_bench_$v.json
  GMMVB_TB_FULL=$v timeout 600 python bench.py --no-cpu --no-legs 2>/dev/null | grep -a "^{" > $OUT/r3X_def_$v.json
  GMMVB_TB_FULL=$v timeout 600 python tools/full_run.py 2>/dev/null | tail -1 > $OUT/r3X_full_$v.json
done
python - <<'PY'
import json
for v in (0,1):
    d=json.load(open("gpurun_out/r3X_bench_%d.json"%v)); e=json.load(open("gpurun_out/r3X_def_%d.json"%v)); f=json.load(open("gpurun_out/r3X_full_%d.json"%v))
    print(v, d["ms_per_step"], d["per_step"]["wall_ms"][:8], d["per_step"]["proof_pairs_per_sample"][:8], "| default", e["ms_per_step"], e["per_step"]["wall_ms"][:4], "| full", f["seconds"])
PY
