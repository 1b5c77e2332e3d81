#!/bin/bash
# usage: tools/sweep_env2.sh <tag> "VAR1=a VAR2=b" "VAR1=c" ...   - one short bench per environment assignment list
TAG=$1; shift
mkdir -p gpurun_out
i=0
for kv in "$@"; do
  i=$((i+1))
  env $kv timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > gpurun_out/${TAG}_$i.json 2> gpurun_out/${TAG}_$i.err
  python - <<P
import json
d=json.load(open("gpurun_out/${TAG}_$i.json"))
ps=d["per_step"]
print("$kv", "ms/step %.2f" % d["ms_per_step"])
print("  E", ps["estep_ms"]); print("  M", ps["mstep_ms"]); print("  eval", ps["evaluated_components_per_sample"]); print("  settled", ps["settled_rows_per_sample"])
P
done
