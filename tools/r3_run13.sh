#!/bin/bash
# lazy sweep: parity subset, then the driver's bench with and without it
set -u
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_sparse_parity.py tests/test_gpu_proof.py -x -q -m gpu 2>&1 | tail -15 > $OUT/r3m_tests.log; tail -5 $OUT/r3m_tests.log
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3m_bench_lazy.json 2> $OUT/r3m_bench_lazy.err; tail -c 200 $OUT/r3m_bench_lazy.err
GMMVB_SWEEP_LAZY=0 timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3m_bench_nolazy.json 2> $OUT/r3m_bench_nolazy.err
python - <<'PY'
import json
for f in ("lazy","nolazy"):
    try:
        d=json.load(open("gpurun_out/r3m_bench_%s.json"%f))
        print(f, d["ms_per_step"], d["per_step"]["wall_ms"], d["per_step"]["estep_ms"], d["roofline"]["pairs_per_sample"])
    except Exception as e:
        print(f, "failed", e)
PY
