#!/bin/bash
# Kernel-trace A/B of variant library builds on the HMM benchmark: tools/ab_hmm_trace.sh <tag> <kernel substrings, comma separated> <variant|-> ...
# prints, per variant, the mean duration of the kernels whose names contain one of the substrings
set -u
TAG=$1; PAT=$2; shift 2
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for v in "$@"; do
  if [ "$v" = "-" ]; then unset BAYESML_AMD_LIB; else export BAYESML_AMD_LIB=$GRAFT_REPO_ROOT/bayesml_amd/csrc/libgmmvb_$v.so; fi
  rm -rf $OUT/${TAG}_abtrace_$v
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_abtrace_$v -- python3 $GRAFT_REPO_ROOT/tools/bench_hmm.py --no-cpu --no-viterbi --steps 5 --warmup 2 > $OUT/${TAG}_abtrace_$v.json 2> $OUT/${TAG}_abtrace_$v.err)
  python3 - $OUT/${TAG}_abtrace_$v "$PAT" "$v" $OUT/${TAG}_abtrace_$v.json <<'PY'
import csv, glob, json, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
pats = sys.argv[2].split(",")
acc = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void gmmvb::", "").split("(")[0]
    if any(p in n for p in pats):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d > 0.02:                # (kernels that returned at a shut gate do not count)
            acc.setdefault(n, []).append(d)
line = json.load(open(sys.argv[4]))
print("[%s]" % sys.argv[3], "ms/iteration", round(line["ms_per_step"], 3), {k: (len(v), round(sum(v) / len(v), 3)) for k, v in acc.items()})
PY
  find $OUT/${TAG}_abtrace_$v -name "*.csv" -delete
done
unset BAYESML_AMD_LIB
