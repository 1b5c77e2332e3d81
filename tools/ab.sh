#!/bin/bash
# A/B measurements on ONE GPU box (inside a gpurun call, from the repo root).  One script for what used to be
# ab_env.sh / ab_lib.sh / ab_hmm.sh / ab_hmm_trace.sh / sweep_env{,2,3}.sh / pmc_ab.sh / pmc_kernel.sh / trace_c4.sh:
#
#   tools/ab.sh env   <tag> <bench args | -> "<ENV=val ...>" ...     bench.py under library switches ("-" = none); every
#                                                                     variant twice, interleaved; ms/step and kernel groups
#   tools/ab.sh lib   <tag> <bench args | -> <variant|-> ...         variant builds (tools/build_variant.sh, BAYESML_AMD_LIB)
#   tools/ab.sh steps <tag> <bench args | -> "<ENV=val ...>" ...     like env, once, printing every 5th step's phases
#   tools/ab.sh hmm   <tag> "<ENV=val ...>" ...                       tools/bench_hmm.py under switches (BAYESML_AMD_LIB=... for builds)
#   tools/ab.sh hmmtrace <tag> <kernel substrings,comma> <variant|-> ...   rocprofv3 kernel trace of bench_hmm.py per build: mean ms
#                                                                     of the kernels named
#   tools/ab.sh trace <tag> <bench args | ->                          rocprofv3 --kernel-trace --stats of bench.py: summary + three steps
#   tools/ab.sh pmc   <tag> <counters,comma> <bench args | ->         one rocprofv3 --pmc pass of bench.py: per-kernel counter means
#
# bench args default to the driver's command (--steps 20 --warmup 5); outputs under gpurun_out/<tag>_*.
set -u
MODE=$1; TAG=$2; shift 2
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
DEFAULT_ARGS="--steps 20 --warmup 5"

bench_args() { if [ "$1" = "-" ]; then echo "$DEFAULT_ARGS"; else echo "$1"; fi; }

report_bench() {      # <detail json> <label>
  python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
g = d["roofline"]["kernel_groups"]
p = d["roofline"]["pairs_per_sample"]
w = d["per_step"]["wall_ms"]
print("[%s]" % sys.argv[2], "ms/step", round(d["ms_per_step"], 3), {k: round(v["ms"], 3) for k, v in g.items()},
      "outside", round(d["roofline"]["outside_events_ms_per_step"], 3), "pairs", {k: round(v, 2) for k, v in p.items()},
      "first/last", w[0], w[-1])
PY
}

case $MODE in
  env|lib)
    ARGS=$(bench_args "$1"); shift
    for rep in 1 2; do
      i=0
      for v in "$@"; do
        i=$((i+1))
        unset BAYESML_AMD_LIB
        if [ $MODE = lib ]; then
          [ "$v" != "-" ] && export BAYESML_AMD_LIB=$ROOT/bayesml_amd/csrc/libgmmvb_$v.so
          envs=""
        else
          envs=$v; [ "$v" = "-" ] && envs=""
        fi
        env $envs timeout 900 python3 bench.py --no-cpu --no-legs --detail $OUT/${TAG}_ab_$i.json $ARGS > /dev/null 2> $OUT/${TAG}_ab_$i.err
        report_bench $OUT/${TAG}_ab_$i.json "$v"
      done
    done
    unset BAYESML_AMD_LIB ;;
  steps)
    ARGS=$(bench_args "$1"); shift
    i=0
    for v in "$@"; do
      i=$((i+1)); envs=$v; [ "$v" = "-" ] && envs=""
      env $envs timeout 900 python3 bench.py --no-cpu --no-legs --detail $OUT/${TAG}_steps_$i.json $ARGS > /dev/null 2> $OUT/${TAG}_steps_$i.err
      python3 - $OUT/${TAG}_steps_$i.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); ps = d["per_step"]
print("[%s]" % sys.argv[2], "ms/step %.2f" % d["ms_per_step"])
for k in ("wall_ms", "estep_ms", "mstep_ms", "evaluated_components_per_sample", "settled_rows_per_sample", "proof_pairs_per_sample", "estep_kernel"):
    print("  ", k, ps[k][::5])
PY
    done ;;
  hmm)
    for rep in 1 2; do
      i=0
      for v in "$@"; do
        i=$((i+1)); envs=$v; [ "$v" = "-" ] && envs=""
        env $envs timeout 600 python3 tools/bench_hmm.py --no-cpu --steps 5 --warmup 2 2>/dev/null | grep -a "^{" > $OUT/${TAG}_hmm_ab_$i.json
        python3 - $OUT/${TAG}_hmm_ab_$i.json "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("[%s]" % sys.argv[2], "ms/iteration", round(d["ms_per_step"], 3), d["boundary_pass"], "viterbi", round(d["viterbi"]["ms"], 2), "vl", d["final_vl"])
PY
      done
    done ;;
  hmmtrace)
    PAT=$1; shift
    for v in "$@"; do
      unset BAYESML_AMD_LIB
      [ "$v" != "-" ] && export BAYESML_AMD_LIB=$ROOT/bayesml_amd/csrc/libgmmvb_$v.so
      rm -rf $ROOT/$OUT/${TAG}_abtrace_$v
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/${TAG}_abtrace_$v -- python3 $ROOT/tools/bench_hmm.py --no-cpu --no-viterbi --steps 5 --warmup 2 > $ROOT/$OUT/${TAG}_abtrace_$v.json 2> $ROOT/$OUT/${TAG}_abtrace_$v.err)
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
    unset BAYESML_AMD_LIB ;;
  trace)
    ARGS=$(bench_args "$1")
    rm -rf $ROOT/$OUT/${TAG}_trace
    (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/${TAG}_trace -- python3 $ROOT/bench.py --no-cpu --no-legs --detail $ROOT/$OUT/${TAG}_bench_detail_profiled.json $ARGS > $ROOT/$OUT/${TAG}_bench_line_profiled.json 2> $ROOT/$OUT/${TAG}_trace.err)
    python3 tools/summarize_rocprof.py $OUT/${TAG}_trace > $OUT/${TAG}_bench_kernel_summary.md 2>> $OUT/${TAG}_trace.err
    head -40 $OUT/${TAG}_bench_kernel_summary.md
    f=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python3 tools/trace_steps.py $f 7 12 24 > $OUT/${TAG}_bench_steps.txt 2>&1
    find $OUT/${TAG}_trace -name "*kernel_trace.csv" -size +20M -delete ;;
  pmc)
    CTR=$(echo $1 | tr ',' ' '); ARGS=$(bench_args "$2")
    rm -rf $ROOT/$OUT/${TAG}_pmcx
    (cd /tmp && timeout 900 rocprofv3 --pmc $CTR --output-format csv -d $ROOT/$OUT/${TAG}_pmcx -- python3 $ROOT/bench.py --no-cpu --no-legs --detail "" $ARGS > /dev/null 2> $ROOT/$OUT/${TAG}_pmcx.err)
    python3 - $OUT/${TAG}_pmcx <<'PY'
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if n.startswith("gmmvb::"):
            rows[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(rows.items(), key=lambda kv: -sum(next(iter(kv[1].values())))):
    print(n[:70], {k: "%.3g" % (sum(v) / len(v)) for k, v in sorted(c.items())}, "n=%d" % len(next(iter(c.values()))))
PY
    find $OUT/${TAG}_pmcx -name "*.csv" -size +20M -delete ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
