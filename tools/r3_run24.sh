#!/bin/bash
set -u
OUT=gpurun_out
for v in default all; do
  if [ $v = all ]; then export GMMVB_PROOF=all; else unset GMMVB_PROOF; fi
  timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3y_bench_$v.json 2> /dev/null
  timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3y_c4_$v.json 2> /dev/null
done
python - <<'PY'
import json
for v in ("default","all"):
    for f in ("bench","c4"):
        d=json.load(open("gpurun_out/r3y_%s_%s.json"%(f,v)))
        p=d["roofline"]["pairs_per_sample"]
        print(f, v, round(d["ms_per_step"],2), "eval", round(p["evaluated_exactly"],2), "proof", round(p["proof_round_int8"],2), "exits", round(p["early_exits"],2))
PY
