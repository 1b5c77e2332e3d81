#!/bin/bash
# A/B of one library switch on the same box: tools/ab_env.sh <tag> <ENV_NAME> <value A> <value B> [bench args...]
# (the driver's command by default: --steps 20 --warmup 5); prints ms per step, E / M phase means and the M-step group
set -u
TAG=$1; NAME=$2; A=$3; B=$4; shift 4
ARGS=${@:-"--steps 20 --warmup 5"}
OUT=gpurun_out
mkdir -p $OUT
for v in "$A" "$B" "$A" "$B"; do
  env $NAME=$v timeout 600 python bench.py --no-cpu --no-legs --detail - $ARGS 2>/dev/null | grep -a "^{" | head -1 > $OUT/${TAG}_${NAME}_$v.json
  python - $OUT/${TAG}_${NAME}_$v.json $NAME $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
g = d["roofline"]["kernel_groups"]
print(sys.argv[2], "=", sys.argv[3], "ms/step", round(d["ms_per_step"], 3), "E", round(d["roofline"]["phase_ms"]["estep"], 3), "M",
      round(d["roofline"]["phase_ms"]["mstep"], 3), "mstep_main", round(g.get("mstep_main", {}).get("ms", 0), 3),
      "tflops", round(g.get("mstep_main", {}).get("executed_f64_tflops", 0), 1), "gather", round(g.get("estep_gather", {}).get("ms", 0), 3), "proof", round(g.get("estep_proof", {}).get("ms", 0), 3), "select", round(g.get("estep_select", {}).get("ms", 0), 3),
      "launch", d["launch"].split("|")[-1].strip()[:40])
PY
done
