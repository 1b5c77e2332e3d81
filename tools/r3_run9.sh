#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > $OUT/r3i_gpu_tests.log 2>&1; tail -15 $OUT/r3i_gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
