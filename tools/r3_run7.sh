#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_generic.py tests/test_mvn.py tests/test_gpu_edges.py -x -q -m gpu > $OUT/r3g_tests1.log 2>&1; tail -15 $OUT/r3g_tests1.log
