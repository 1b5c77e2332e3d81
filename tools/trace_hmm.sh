#!/bin/bash
# rocprofv3 kernel trace of tools/bench_hmm.py; summary to gpurun_out/<tag>_hmm_kernel_summary.md
TAG=$1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/${TAG}_hmm_trace
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_hmm_trace -- python3 $GRAFT_REPO_ROOT/tools/bench_hmm.py --no-cpu > $OUT/${TAG}_hmm_line_profiled.json 2> $OUT/${TAG}_hmm_trace.err)
python3 tools/summarize_rocprof.py $OUT/${TAG}_hmm_trace > $OUT/${TAG}_hmm_kernel_summary.md 2>> $OUT/${TAG}_hmm_trace.err
head -30 $OUT/${TAG}_hmm_kernel_summary.md
find $OUT/${TAG}_hmm_trace -name "*kernel_trace.csv" -size +20M -delete
