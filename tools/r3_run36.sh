#!/bin/bash
set -u
OUT=gpurun_out
for v in 4 32; do
  GMMVB_REGROUP_ACT=$v timeout 600 python bench.py --no-cpu --no-legs 2>/dev/null | grep -a "^{" > $OUT/r3a2_def_$v.json
  GMMVB_REGROUP_ACT=$v timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3a2_w5_$v.json
  GMMVB_REGROUP_ACT=$v timeout 600 python tools/full_run.py 2>/dev/null | tail -1 > $OUT/r3a2_full_$v.json
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
e=json.load(open("gpurun_out/r3a2_def_%s.json"%v)); d=json.load(open("gpurun_out/r3a2_w5_%s.json"%v)); f=json.load(open("gpurun_out/r3a2_full_%s.json"%v))
print(v, "default", round(e["ms_per_step"],2), e["per_step"]["wall_ms"], [k[6:12] for k in e["per_step"]["estep_kernel"]][:5], "| w5s20", round(d["ms_per_step"],2), d["per_step"]["wall_ms"][:4], "| full", round(f["seconds"],3))
PY
done
