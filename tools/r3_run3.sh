#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_tiled.py tests/test_gpu_proof.py -x -q -m gpu > $OUT/r3c_tests1.log 2>&1; tail -12 $OUT/r3c_tests1.log
for cfg in "m0:" "m5:GMMVB_SETTLE_MARGIN=5" "m2:GMMVB_SETTLE_MARGIN=2" "all:GMMVB_PROOF=all"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 600 python bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $OUT/r3c_bench_$tag.json 2> $OUT/r3c_bench_$tag.err; tail -c 300 $OUT/r3c_bench_$tag.err
done
python - <<'PY'
import json
for n in ("m0","m5","m2","all"):
    try:
        d=json.load(open(f"gpurun_out/r3c_bench_{n}.json"))
        print(n, round(d["ms_per_step"],3), d["roofline"]["pairs_per_sample"], {k:round(v["ms"],2) for k,v in d["roofline"]["kernel_groups"].items()})
        for w in d["warmup_steps"]: print("   warm", w["kernels"][0], w["estep_ms"], w["mstep_ms"], w["active_components_per_sample"], w["evaluated_components_per_sample"])
        p=d["per_step"]
        for k in ("wall_ms","estep_ms","evaluated_components_per_sample","settled_rows_per_sample","proof_pairs_per_sample"): print("  ",k,p[k])
        print("   kern", [k[6:12] for k in p["estep_kernel"]])
    except Exception as e: print(n, "failed", e)
PY
ROWS=2000000 timeout 600 python tools/drift_anchor.py > $OUT/r3c_drift_anchor.json 2> $OUT/r3c_drift_anchor.err; tail -c 300 $OUT/r3c_drift_anchor.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3c_drift_anchor.json"))
for a,rows in d.items():
    print(a)
    for r in rows: print("  t",r["t"],"direct min/med",round(r["direct_min"],3),round(r["direct_med"],4),"product",round(r["product_min"],3),round(r["product_med"],4), r["worst5_direct"], r["worst5_product"])
PY
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_sparse_parity.py tests/test_gpu_sharded.py -x -q -m gpu > $OUT/r3c_tests2.log 2>&1; tail -15 $OUT/r3c_tests2.log
timeout 900 python bench.py --config c4 --scaling strong --gpus 1 --no-cpu --no-legs --steps 3 --warmup 2 > $OUT/r3c_bench_c4_strong1.json 2> $OUT/r3c_c4s.err; tail -c 600 $OUT/r3c_c4s.err; head -c 1500 $OUT/r3c_bench_c4_strong1.json
