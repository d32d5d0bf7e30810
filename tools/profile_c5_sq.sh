#!/bin/bash
# SQ counters of the bf16 configs[4] step (separate --pmc passes, no trace domains): matrix-pipe busy, waits, LDS activity and conflicts
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/c5_sq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $O/p$i -o p -- python $R/tools/bench_bf16.py c5bf16 > /dev/null 2> $O/p$i.err
done
python $R/tools/pmc_any.py $O/sq.json $O/p1 $O/p2 $O/p3 > $O/sq.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
grep -A1 "bf16" $O/sq.txt | head -40
