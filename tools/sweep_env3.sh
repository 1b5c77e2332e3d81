#!/bin/bash
# usage: tools/sweep_env3.sh <tag> <steps> "VAR1=a VAR2=b" ...   - like sweep_env2.sh with a step count; prints every 5th step
TAG=$1; STEPS=$2; shift 2
mkdir -p gpurun_out
i=0
for kv in "$@"; do
  i=$((i+1))
  env $kv timeout 900 python bench.py --no-cpu --no-legs --steps $STEPS --warmup 5 > gpurun_out/${TAG}_$i.json 2> gpurun_out/${TAG}_$i.err
  python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_$i.json"))
ps=d["per_step"]
print("$kv", "ms/step %.2f" % d["ms_per_step"])
for k in ("estep_ms","mstep_ms","evaluated_components_per_sample","settled_rows_per_sample","estep_kernel"):
    print("  ", k, ps[k][::5])
P
done
