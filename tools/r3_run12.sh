#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/r3l_pmc; (cd /tmp && timeout 900 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "mstep_list|estep_gather|rec_sweep|fill_lists|rec_finish|estep_i8|rec_build|drift_kernel|kside_step" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3l_pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/$OUT/r3l_pmc.json 2> $GRAFT_REPO_ROOT/$OUT/r3l_pmc.err); tail -c 300 $OUT/r3l_pmc.err
python tools/summarize_pmc.py $OUT/r3l_pmc > $OUT/r3l_pmc_summary.md 2>&1; cat $OUT/r3l_pmc_summary.md | head -150
find $OUT/r3l_pmc -name "*.csv" -size +20M -delete
