#!/bin/bash
set -u
OUT=gpurun_out
for v in "8 6" "11 6" "11 9" "13 6"; do
  set -- $v
  GMMVB_DRIFT_SQ=$1 GMMVB_DRIFT_SQ_BIG=$2 timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/r3a1_bench_$1_$2.json
  python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r3a1_bench_%s_%s.json"%(sys.argv[1],sys.argv[2])))
p=d["roofline"]["pairs_per_sample"]
print(sys.argv[1:], round(d["ms_per_step"],3), "proof", round(p["proof_round_int8"],3), "eval", round(p["evaluated_exactly"],3), "outside", round(d["roofline"]["outside_events_ms_per_step"],2), d["per_step"]["wall_ms"][1:6])
PY
done
