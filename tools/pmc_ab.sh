#!/bin/bash
# PMC FETCH_SIZE of one bench command under two values of a library switch: tools/pmc_ab.sh <tag> <ENV_NAME> <A> <B> <kernel substring> [bench args...]
# (the switch is exported BEFORE rocprofv3 starts: the program after `--` is python3 itself)
set -u
TAG=$1; NAME=$2; A=$3; B=$4; KER=$5; shift 5
ARGS=${@:-"--steps 20 --warmup 5"}
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for v in "$A" "$B"; do
  export $NAME=$v
  rm -rf $OUT/${TAG}_pmc_$v
  (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_pmc_$v -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs $ARGS > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_pmc_$v.err)
  echo "== $NAME=$v"
  python tools/summarize_pmc.py $OUT/${TAG}_pmc_$v 2>/dev/null | grep -A2 "$KER"
  find $OUT/${TAG}_pmc_$v -name "*.csv" -size +20M -delete
done
