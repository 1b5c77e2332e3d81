#!/bin/bash
# usage: tools/sweep_env.sh <tag> VAR v1 v2 ...   - one short bench per value of an environment knob; prints the last
# five E-step / M-step times of each
TAG=$1; VAR=$2; shift 2
mkdir -p gpurun_out
for v in "$@"; do
  env $VAR=$v timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > gpurun_out/${TAG}_${VAR}_$v.json 2> gpurun_out/${TAG}_${VAR}_$v.err
  python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_${VAR}_$v.json"))
ps=d["per_step"]
print("$VAR=$v", "ms/step %.2f" % d["ms_per_step"], "E", ps["estep_ms"][-5:], "M", ps["mstep_ms"][-5:])
P
done
