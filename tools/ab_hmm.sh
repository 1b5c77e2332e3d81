#!/bin/bash
# A/B of library switches on the HMM benchmark (tools/bench_hmm.py) on one box: tools/ab_hmm.sh <tag> "<ENV=val ...>" "<ENV=val ...>" ...
# ("-" = no switch); every variant runs twice, interleaved
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
for rep in 1 2; do
  i=0
  for v in "$@"; do
    i=$((i+1))
    [ "$v" = "-" ] && v=""
    env $v timeout 600 python tools/bench_hmm.py --no-cpu --steps 5 --warmup 2 2>/dev/null | grep -a "^{" > $OUT/${TAG}_hmm_ab_$i.json
    python - $OUT/${TAG}_hmm_ab_$i.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("[%s]" % sys.argv[2], "ms/iteration", round(d["ms_per_step"], 3), d["boundary_pass"], "viterbi", round(d["viterbi"]["ms"], 2), "vl", d["final_vl"])
PY
  done
done
