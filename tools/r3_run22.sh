#!/bin/bash
set -u
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); rm -rf $OUT/r3w_pmc$i
  (cd /tmp && timeout 900 rocprofv3 --pmc $set --kernel-include-regex "rec_sweep|rec_finish|fill_lists|rec_proof_decide" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3w_pmc$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-legs --steps 20 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/r3w_pmc$i.err)
  tail -c 200 $OUT/r3w_pmc$i.err
done
python tools/summarize_pmc.py $OUT/r3w_pmc1 $OUT/r3w_pmc2 > $OUT/r3w_pmc_summary.md 2>&1; head -80 $OUT/r3w_pmc_summary.md
find $OUT/r3w_pmc1 $OUT/r3w_pmc2 -name "*.csv" -size +20M -delete
