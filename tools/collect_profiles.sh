#!/bin/bash
# Copy the summaries of one tools/gpu_round.sh run (gpurun_out/<tag>_*) into profiles/<round>_* (tracked).
# usage: tools/collect_profiles.sh r3A r3
set -e
tag=$1; rnd=$2
cd "$(dirname "$0")/.."
for f in bench_line.json bench_line_w5s20.json bench_line_c2.json bench_line_c4.json bench_line_profiled.json \
         bench_line_force_dist_native.json bench_line_force_dist_torch.json bench_kernel_summary.md bench_kernel_stats.csv \
         pmc_summary.md small.json full_run.json hmm_bench_line.json hmm_kernel_summary.md gpu_tests.txt \
         bench_line_c4_strong1.json bench_line_c4_w5s20.json bench_proof.json hmm_pmc_summary.md hmm_k128_line.json \
         bench_steps.txt bench_line_spread1.json active_hist.json \
         bench_line_eight_ranks_one_gpu.json hmm_k128_t2e6_line.json hmm_k256_t5e5_line.json c4_kernel_summary.md c4_steps.txt bench_line_c4_profiled.json bench_line_all_legs.json bench_detail_all_legs.json bench_detail_w5s20.json bench_detail_c4_w5s20.json bench_line_c4_strong_eight_ranks_one_gpu.json hmm_t10000_graph1.json hmm_t10000_graph0.json hmm_t100000_graph1.json hmm_t100000_graph0.json bench_line_dense_k32_d128.json bench_line_dense_k32_d160.json bench_line_dense_k32_d200.json bench_line_dense_k32_d256.json; do
    [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${rnd}_$f
    case $f in *.json) [ -f profiles/${rnd}_$f ] && grep -a '^{' profiles/${rnd}_$f > profiles/${rnd}_$f.tmp && mv profiles/${rnd}_$f.tmp profiles/${rnd}_$f ;; esac   # (RCCL prints a banner on stdout)
done
[ -f gpurun_out/${tag}_pmc_traffic.json ] && cp gpurun_out/${tag}_pmc_traffic.json profiles/pmc_traffic.json
[ -f gpurun_out/${tag}_hmm_pmc_traffic.json ] && cp gpurun_out/${tag}_hmm_pmc_traffic.json profiles/hmm_pmc_traffic.json
[ -f gpurun_out/${tag}_gpu_tests.log ] && tail -8 gpurun_out/${tag}_gpu_tests.log > profiles/${rnd}_gpu_tests.txt
ls -la profiles/${rnd}_* profiles/pmc_traffic.json
