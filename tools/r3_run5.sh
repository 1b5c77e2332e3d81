#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_hmm.py tests/test_gpu_sparse_parity.py tests/test_gpu_proof.py -x -q -m gpu > $OUT/r3e_tests1.log 2>&1; tail -12 $OUT/r3e_tests1.log
timeout 300 python tools/bench_proof.py > $OUT/r3e_bench_proof.json 2> $OUT/r3e_bench_proof.err; cat $OUT/r3e_bench_proof.json
timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3e_bench.json 2> $OUT/r3e_bench.err; tail -c 300 $OUT/r3e_bench.err
python - <<'PY'
import json
for n in ("",):
    try:
        d=json.load(open(f"gpurun_out/r3e_bench{n}.json"))
        print(n, round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
        for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
        p=d["per_step"]
        for k in ("wall_ms","estep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
    except Exception as e: print(n, "failed", e)
PY
timeout 900 python tools/bench_hmm.py --no-cpu > $OUT/r3e_hmm.json 2> $OUT/r3e_hmm.err; tail -c 400 $OUT/r3e_hmm.err; head -c 1500 $OUT/r3e_hmm.json; echo
timeout 900 python bench.py --config c4 --scaling strong --gpus 1 --no-cpu --no-legs --steps 3 --warmup 2 > $OUT/r3e_bench_c4_strong1.json 2> $OUT/r3e_c4s.err; tail -c 800 $OUT/r3e_c4s.err; head -c 1200 $OUT/r3e_bench_c4_strong1.json; echo
timeout 900 python bench.py --config c4 --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3e_bench_c4.json 2> $OUT/r3e_c4.err; tail -c 300 $OUT/r3e_c4.err
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/r3e_bench_c4.json"))
    print("c4", round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
    p=d["per_step"]
    for k in ("wall_ms","estep_ms","mstep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
    print("   kern", [k[6:12] for k in p["estep_kernel"]])
except Exception as e: print("c4 failed", e)
PY
