#!/bin/bash
# whole-fit seconds (bench.py's full_fit leg only) under library switches: tools/full_fit_ab.sh <tag> "<ENV=val ...>" ...
set -u
TAG=$1; shift
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1)); envs=$v; [ "$v" = "-" ] && envs=""
    env $envs timeout 900 python3 bench.py --no-cpu --legs full --steps 2 --warmup 1 --detail $OUT/${TAG}_fullfit_$i.json > /dev/null 2> $OUT/${TAG}_fullfit_$i.err
    python3 - $OUT/${TAG}_fullfit_$i.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["full_fit"]
print("[%s]" % sys.argv[2], "seconds", round(d["seconds"], 4), d["kernel_launches"], "vl", d["final_vl"])
PY
  done
done
