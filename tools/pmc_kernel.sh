#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <counters...>   - one rocprofv3 --pmc pass of the bench (no legs), per-kernel averages
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmcx
rm -rf $OUT
(cd /tmp && timeout 900 rocprofv3 --pmc "$@" --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > /dev/null 2> $OUT.err)
python3 - <<P
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if n.startswith("gmmvb::"):
            rows[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in sorted(rows.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    print(n[:70], {k: "%.3g" % (sum(v) / len(v)) for k, v in sorted(c.items())}, "n=%d" % len(next(iter(c.values()))))
P
find $OUT -name "*.csv" -size +20M -delete
