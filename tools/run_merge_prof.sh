#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/merge; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in ${MODES:-1}; do
  BP_MERGE=$m rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$m -o kt -- python $R/tools/bench_bf16.py c2f32 > $O/kt$m.log 2>&1
  echo "== BP_MERGE=$m"; tail -1 $O/kt$m.log; cut -d, -f1-4 $(find $O/kt$m -name '*kernel_stats.csv') | head -12
done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
