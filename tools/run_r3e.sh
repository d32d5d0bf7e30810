#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5bf16 > /dev/null 2> $O/pmc$i.err
done
python $GRAFT_REPO_ROOT/tools/pmc_any.py $O/c5_sq.json $O/pmc1 $O/pmc2 $O/pmc3 | grep -A1 "bp_gemm_bf16<0, 128>\|bp_gemm_bf16<2, 128>\|bp_wgrad_dma_bf16"
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
