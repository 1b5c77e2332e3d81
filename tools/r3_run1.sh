#!/bin/bash
# round 3, first GPU call: proof kernel tests + micro-benchmark, sparse parity suite, bench with / without the proof round
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_proof.py -x -q -m gpu > $OUT/r3a_proof_tests.log 2>&1; tail -15 $OUT/r3a_proof_tests.log
timeout 300 python tools/bench_proof.py > $OUT/r3a_bench_proof.json 2> $OUT/r3a_bench_proof.err; tail -c 300 $OUT/r3a_bench_proof.err; cat $OUT/r3a_bench_proof.json
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3a_bench_proof_on.json 2> $OUT/r3a_bench_on.err; tail -c 400 $OUT/r3a_bench_on.err
GMMVB_PROOF=0 timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3a_bench_proof_off.json 2> $OUT/r3a_bench_off.err; tail -c 400 $OUT/r3a_bench_off.err
python - <<'PY'
import json
for n in ("on","off"):
    try:
        d=json.load(open(f"gpurun_out/r3a_bench_proof_{n}.json"))
        print(n, round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
        print("  wall", d["per_step"]["wall_ms"]); print("  E", d["per_step"]["estep_ms"]); print("  M", d["per_step"]["mstep_ms"]); print("  eval", d["per_step"]["evaluated_components_per_sample"]); print("  settled", d["per_step"]["settled_rows_per_sample"])
    except Exception as e: print(n, "failed", e)
PY
timeout 2400 python -m pytest tests/test_gpu_sparse_parity.py tests/test_samplers.py tests/test_gpu_sharded.py -x -q -m gpu > $OUT/r3a_tests.log 2>&1; tail -25 $OUT/r3a_tests.log
