#!/bin/bash
# A/B of a variant library build (tools/build_variant.sh) against the tree's own on the same box:
#   tools/ab_lib.sh <tag> <variant name> [bench args...]     (the driver's command by default)
set -u
TAG=$1; VAR=$2; shift 2
ARGS=${@:-"--steps 20 --warmup 5"}
OUT=gpurun_out
mkdir -p $OUT
for v in base $VAR base $VAR; do
  if [ $v = base ]; then unset BAYESML_AMD_LIB; else export BAYESML_AMD_LIB=$PWD/bayesml_amd/csrc/libgmmvb_$v.so; fi
  timeout 600 python bench.py --no-cpu --no-legs --detail - $ARGS 2>/dev/null | grep -a "^{" | head -1 > $OUT/${TAG}_$v.json
  python - $OUT/${TAG}_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
g = d["roofline"]["kernel_groups"]
p = d["roofline"]["pairs_per_sample"]
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), {k: round(v["ms"], 2) for k, v in g.items()},
      "pairs", {k: round(v, 2) for k, v in p.items()}, "first/last", d["per_step"]["wall_ms"][0], d["per_step"]["wall_ms"][-1])
PY
done
unset BAYESML_AMD_LIB
