#!/bin/bash
set -u
OUT=gpurun_out
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf $OUT/r3F_pmc$i
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-include-regex "estep_i8_proof" --output-format csv -d $GRAFT_REPO_ROOT/$OUT/r3F_pmc$i -- python3 $GRAFT_REPO_ROOT/tools/bench_proof.py > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/r3F_pmc$i.err)
  tail -c 150 $OUT/r3F_pmc$i.err
done
python tools/summarize_pmc.py $OUT/r3F_pmc1 $OUT/r3F_pmc2 $OUT/r3F_pmc3 > $OUT/r3F_pmc_summary.md 2>&1; cat $OUT/r3F_pmc_summary.md
