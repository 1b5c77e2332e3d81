#!/bin/bash
# Mean duration of the kernels named (comma-separated substrings) in a rocprofv3 kernel trace of bench.py, per library build:
#   tools/trace_kernels_ab.sh <tag> <substrings> <bench args | -> <variant|-> ...
set -u
TAG=$1; PAT=$2; ARGS=$3; shift 3
[ "$ARGS" = "-" ] && ARGS="--steps 20 --warmup 5"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  unset BAYESML_AMD_LIB
  [ "$v" != "-" ] && export BAYESML_AMD_LIB=$ROOT/bayesml_amd/csrc/libgmmvb_$v.so
  rm -rf $ROOT/$OUT/${TAG}_kab_$v
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/${TAG}_kab_$v -- python3 $ROOT/bench.py --no-cpu --no-legs --detail "" $ARGS > /dev/null 2> $ROOT/$OUT/${TAG}_kab_$v.err)
  python3 - $OUT/${TAG}_kab_$v "$PAT" "$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pats = sys.argv[2].split(",")
acc = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void gmmvb::", "").split("(")[0].split("<")[0]
    if any(p in n for p in pats):
        acc.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("[%s]" % sys.argv[3], {k: (len(v), "mean %.1f us" % (sum(v) / len(v)), "median %.1f" % sorted(v)[len(v) // 2]) for k, v in sorted(acc.items())})
PY
  find $OUT/${TAG}_kab_$v -name "*.csv" -delete
done
unset BAYESML_AMD_LIB
