#!/bin/bash
# usage: tools/collect_blend_pmc.sh <tag> [bench args]  -- the SQ counter passes VERDICT r02 item 5 asks for (issue / wait / LDS), per kernel
tag=$1; shift
for pass in "A:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES" \
            "B:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  bash tools/pmc_brief.sh ${tag}_$name "$ctrs" "$@" | grep -E "kernel|blend_|preprocess_|run_scatter|emit_column|depth_bucket" | cut -c1-260 > gpurun_out/pmcsq_${tag}_$name.txt
  cat gpurun_out/pmcsq_${tag}_$name.txt
done
