#!/bin/bash
# One gpurun call: tests, bench legs, rocprofv3 kernel trace and PMC passes.  usage: tools/gpu_round.sh <tag> <stage>...
# (bench.py prints its compact line on stdout; every stage keeps the full record next to it as *_detail.json)
# stages: smoke tests newtests c2trace c4trace c4strong1 bench bench20 dense c2 c4 c4w5 c4strong trace steps tracedel pmc hmm hmmtrace hmmpmc hmmbig full small dist proofbench spread1 hist wide share8
# (pmc / hmmpmc first: the bench stages quote the traffic files they write)
# Outputs under gpurun_out/<tag>_*; copy the summaries worth keeping into profiles/.
set -u
TAG=$1; shift
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
for stage in "$@"; do
  case $stage in
    newtests) timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py -x -q -m gpu > $OUT/${TAG}_newtests.log 2>&1; tail -15 $OUT/${TAG}_newtests.log ;;
    tests) timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/${TAG}_gpu_tests.log 2>&1; tail -8 $OUT/${TAG}_gpu_tests.log ;;
    bench) timeout 900 python bench.py --detail $OUT/${TAG}_bench_line_detail.json > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench.err; tail -c 600 $OUT/${TAG}_bench.err; head -c 1500 $OUT/${TAG}_bench_line.json; echo ;;
    bench20) timeout 900 python bench.py --steps 20 --warmup 5 --detail $OUT/${TAG}_bench_detail_w5s20.json > $OUT/${TAG}_bench_line_w5s20.json 2> $OUT/${TAG}_bench20.err; tail -c 600 $OUT/${TAG}_bench20.err; cat $OUT/${TAG}_bench_line_w5s20.json; echo ;;
    dense) timeout 600 python bench.py --dense --no-cpu --detail $OUT/${TAG}_bench_line_dense_detail.json > $OUT/${TAG}_bench_line_dense.json 2> $OUT/${TAG}_dense.err; head -c 600 $OUT/${TAG}_bench_line_dense.json; echo ;;
    c4) timeout 900 python bench.py --config c4 --no-cpu --detail $OUT/${TAG}_bench_line_c4_detail.json > $OUT/${TAG}_bench_line_c4.json 2> $OUT/${TAG}_c4.err; tail -c 600 $OUT/${TAG}_c4.err; head -c 600 $OUT/${TAG}_bench_line_c4.json; echo ;;
    c2) timeout 600 python bench.py --config c2 --detail $OUT/${TAG}_bench_line_c2_detail.json > $OUT/${TAG}_bench_line_c2.json 2> $OUT/${TAG}_c2.err; tail -c 600 $OUT/${TAG}_c2.err; head -c 600 $OUT/${TAG}_bench_line_c2.json; echo ;;
    steps) f=$(find $OUT/${TAG}_trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_steps.py $f 7 12 24 > $OUT/${TAG}_bench_steps.txt 2>&1; head -12 $OUT/${TAG}_bench_steps.txt ;;
    spread1) timeout 600 python bench.py --no-cpu --no-legs --rows 2000000 --spread 1.0 --steps 20 --warmup 2 --detail $OUT/${TAG}_bench_line_spread1_detail.json > $OUT/${TAG}_bench_line_spread1.json 2> $OUT/${TAG}_spread1.err; head -c 300 $OUT/${TAG}_bench_line_spread1.json; echo ;;
    wide) for d in 128 160 200 256; do timeout 300 python bench.py --dense --no-cpu --no-legs --classes 32 --degree $d --rows 1000000 --steps 3 --warmup 1 --detail $OUT/${TAG}_bench_detail_dense_k32_d$d.json 2>/dev/null | grep -a "^{" > $OUT/${TAG}_bench_line_dense_k32_d$d.json; python - $OUT/${TAG}_bench_detail_dense_k32_d$d.json $d <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); g = d["roofline"]["kernel_groups"]
print("D =", sys.argv[2], "ms/step", round(d["ms_per_step"], 2), {k: (round(v["ms"], 2), round(v.get("executed_f64_tflops", 0), 1)) for k, v in g.items() if v["ms"] > 0.05}, d["launch"][:60])
PY
          done ;;
    share8) BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 8 --rows 200000 --strong-total-rows 1600000 --steps 12 --warmup 5 --no-cpu --legs c4strong --detail $OUT/${TAG}_bench_detail_eight_ranks_one_gpu.json 2> $OUT/${TAG}_share8.err | grep -a "^{" > $OUT/${TAG}_bench_line_eight_ranks_one_gpu.json; tail -c 300 $OUT/${TAG}_share8.err; head -c 700 $OUT/${TAG}_bench_line_eight_ranks_one_gpu.json; echo ;;
    legsall) timeout 1500 python bench.py --steps 20 --warmup 5 --legs all --detail $OUT/${TAG}_bench_detail_all_legs.json > $OUT/${TAG}_bench_line_all_legs.json 2> $OUT/${TAG}_legsall.err; tail -c 300 $OUT/${TAG}_legsall.err; cat $OUT/${TAG}_bench_line_all_legs.json; echo ;;
    strong8) BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 8 --config c4 --scaling strong --total-rows 1600000 --steps 12 --warmup 5 --no-cpu --no-legs --detail $OUT/${TAG}_bench_detail_c4_strong_eight_ranks_one_gpu.json 2> $OUT/${TAG}_strong8.err | grep -a "^{" > $OUT/${TAG}_bench_line_c4_strong_eight_ranks_one_gpu.json; tail -c 300 $OUT/${TAG}_strong8.err; cat $OUT/${TAG}_bench_line_c4_strong_eight_ranks_one_gpu.json; echo ;;
    hmmsmall) for t in 10000 100000; do for g in 1 0; do BAYESML_AMD_KSIDE_GRAPH=$g timeout 300 python tools/bench_hmm.py --rows $t --no-cpu --steps 20 --warmup 5 2>/dev/null | grep -a "^{" > $OUT/${TAG}_hmm_t${t}_graph$g.json; python -c "
import json,sys; d=json.load(open('$OUT/${TAG}_hmm_t${t}_graph$g.json')); print('T=$t graph=$g ms/iteration', round(d['ms_per_step'],3))"; done; done ;;
    hist) timeout 300 python tools/active_hist.py > $OUT/${TAG}_active_hist.json 2> $OUT/${TAG}_hist.err; head -c 300 $OUT/${TAG}_active_hist.json; echo ;;
    trace) rm -rf $OUT/${TAG}_trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 --detail $GRAFT_REPO_ROOT/$OUT/${TAG}_bench_line_profiled_detail.json > $GRAFT_REPO_ROOT/$OUT/${TAG}_bench_line_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_trace.err)
           python tools/summarize_rocprof.py $OUT/${TAG}_trace > $OUT/${TAG}_bench_kernel_summary.md 2>> $OUT/${TAG}_trace.err; head -30 $OUT/${TAG}_bench_kernel_summary.md
           ;;
    c4trace) rm -rf $OUT/${TAG}_c4trace; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_c4trace -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 --detail $GRAFT_REPO_ROOT/$OUT/${TAG}_bench_line_c4_profiled_detail.json > $GRAFT_REPO_ROOT/$OUT/${TAG}_bench_line_c4_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_c4trace.err)
           python tools/summarize_rocprof.py $OUT/${TAG}_c4trace > $OUT/${TAG}_c4_kernel_summary.md 2>> $OUT/${TAG}_c4trace.err; head -40 $OUT/${TAG}_c4_kernel_summary.md
           f=$(find $OUT/${TAG}_c4trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_steps.py $f 7 12 24 > $OUT/${TAG}_c4_steps.txt 2>&1
           find $OUT/${TAG}_c4trace -name "*kernel_trace.csv" -size +20M -delete ;;
    c2trace) rm -rf $OUT/${TAG}_c2trace; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_c2trace -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --no-cpu --no-legs --steps 20 --warmup 5 --detail $GRAFT_REPO_ROOT/$OUT/${TAG}_bench_line_c2_profiled_detail.json > $GRAFT_REPO_ROOT/$OUT/${TAG}_bench_line_c2_profiled.json 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_c2trace.err)
           python tools/summarize_rocprof.py $OUT/${TAG}_c2trace > $OUT/${TAG}_c2_kernel_summary.md 2>> $OUT/${TAG}_c2trace.err; head -24 $OUT/${TAG}_c2_kernel_summary.md
           f=$(find $OUT/${TAG}_c2trace -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_steps.py $f 7 12 24 > $OUT/${TAG}_c2_steps.txt 2>&1; head -30 $OUT/${TAG}_c2_steps.txt
           find $OUT/${TAG}_c2trace -name "*kernel_trace.csv" -size +20M -delete ;;
    c4strong1) timeout 1200 python bench.py --config c4 --scaling strong --gpus 1 --no-cpu --no-legs --steps 20 --warmup 5 --detail $OUT/${TAG}_bench_line_c4_strong1_detail.json > $OUT/${TAG}_bench_line_c4_strong1.json 2> $OUT/${TAG}_c4strong1.err; tail -c 300 $OUT/${TAG}_c4strong1.err; head -c 500 $OUT/${TAG}_bench_line_c4_strong1.json; echo ;;
    tracedel) find $OUT/${TAG}_trace -name "*kernel_trace.csv" -size +20M -delete ;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do rm -rf $OUT/${TAG}_pmc_$c
           (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --detail "" --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_pmc_$c.err); done
         python tools/summarize_pmc.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE --config "K64 D128 N10000000 f32" --window w5s20 --json $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_summary.md 2> $OUT/${TAG}_pmc.err; head -60 $OUT/${TAG}_pmc_summary.md
         [ -s $OUT/${TAG}_pmc_traffic.json ] && cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json                  # (later stages of this call quote it)
         find $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE -name "*.csv" -size +20M -delete ;;
    hmm) timeout 900 python tools/bench_hmm.py > $OUT/${TAG}_hmm_bench_line.json 2> $OUT/${TAG}_hmm.err; tail -c 400 $OUT/${TAG}_hmm.err; head -c 800 $OUT/${TAG}_hmm_bench_line.json; echo ;;
    full) timeout 900 python tools/full_run.py > $OUT/${TAG}_full_run.json 2> $OUT/${TAG}_full.err; tail -c 400 $OUT/${TAG}_full.err; head -c 800 $OUT/${TAG}_full_run.json; echo ;;
    small) timeout 600 python tools/bench_small.py > $OUT/${TAG}_small.json 2> $OUT/${TAG}_small.err; tail -c 300 $OUT/${TAG}_small.err; head -c 900 $OUT/${TAG}_small.json; echo ;;
    dist) timeout 900 python bench.py --force-dist --no-cpu --no-legs --steps 20 --warmup 5 --detail $OUT/${TAG}_bench_line_force_dist_torch_detail.json > $OUT/${TAG}_bench_line_force_dist_torch.json 2> $OUT/${TAG}_dist_torch.err; head -c 400 $OUT/${TAG}_bench_line_force_dist_torch.json; echo
          timeout 900 python bench.py --force-dist --native-allreduce --no-cpu --no-legs --steps 20 --warmup 5 --detail $OUT/${TAG}_bench_line_force_dist_native_detail.json > $OUT/${TAG}_bench_line_force_dist_native.json 2> $OUT/${TAG}_dist_native.err; head -c 400 $OUT/${TAG}_bench_line_force_dist_native.json; echo ;;
    hmmtrace) bash tools/trace_hmm.sh $TAG ;;
    hmmpmc) for c in FETCH_SIZE WRITE_SIZE; do rm -rf $OUT/${TAG}_hmm_pmc_$c
           (cd /tmp && timeout 900 rocprofv3 --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/$OUT/${TAG}_hmm_pmc_$c -- python3 $GRAFT_REPO_ROOT/tools/bench_hmm.py --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_hmm_pmc_$c.err); done
         python tools/hmm_pmc_total.py $OUT/${TAG}_hmm_pmc_FETCH_SIZE $OUT/${TAG}_hmm_pmc_WRITE_SIZE --config "K32 D16 T10000000" --json $OUT/${TAG}_hmm_pmc_traffic.json > $OUT/${TAG}_hmm_pmc_summary.md 2> $OUT/${TAG}_hmm_pmc.err; cat $OUT/${TAG}_hmm_pmc_summary.md
         [ -s $OUT/${TAG}_hmm_pmc_traffic.json ] && cp $OUT/${TAG}_hmm_pmc_traffic.json profiles/hmm_pmc_traffic.json      # (later stages of this call quote it)
         find $OUT/${TAG}_hmm_pmc_FETCH_SIZE $OUT/${TAG}_hmm_pmc_WRITE_SIZE -name "*.csv" -size +20M -delete ;;
    c4w5) timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 --detail $OUT/${TAG}_bench_detail_c4_w5s20.json > $OUT/${TAG}_bench_line_c4_w5s20.json 2> $OUT/${TAG}_c4w5.err; cat $OUT/${TAG}_bench_line_c4_w5s20.json; echo ;;
    c4strong) timeout 1200 python bench.py --config c4 --scaling strong --gpus 1 --no-cpu --no-legs --steps 3 --warmup 1 --detail $OUT/${TAG}_bench_line_c4_strong1_detail.json > $OUT/${TAG}_bench_line_c4_strong1.json 2> $OUT/${TAG}_c4strong.err; tail -c 300 $OUT/${TAG}_c4strong.err; head -c 500 $OUT/${TAG}_bench_line_c4_strong1.json; echo ;;
    proofbench) timeout 600 python tools/bench_proof.py > $OUT/${TAG}_bench_proof.json 2> $OUT/${TAG}_bench_proof.err; head -c 500 $OUT/${TAG}_bench_proof.json; echo ;;
    hmmbig) timeout 900 python tools/bench_hmm.py --classes 128 --degree 8 --rows 200000 --steps 3 --warmup 1 --no-cpu > $OUT/${TAG}_hmm_k128_line.json 2> $OUT/${TAG}_hmmbig.err; head -c 300 $OUT/${TAG}_hmm_k128_line.json; echo
            timeout 900 python tools/bench_hmm.py --classes 128 --degree 8 --rows 2000000 --steps 3 --warmup 1 --no-cpu --no-viterbi > $OUT/${TAG}_hmm_k128_t2e6_line.json 2>> $OUT/${TAG}_hmmbig.err; head -c 300 $OUT/${TAG}_hmm_k128_t2e6_line.json; echo
            timeout 900 python tools/bench_hmm.py --classes 256 --degree 8 --rows 500000 --steps 2 --warmup 1 --no-cpu --no-viterbi > $OUT/${TAG}_hmm_k256_t5e5_line.json 2>> $OUT/${TAG}_hmmbig.err; head -c 300 $OUT/${TAG}_hmm_k256_t5e5_line.json; echo ;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ;;
    *) echo "unknown stage $stage" ;;
  esac
done
