#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/bfdma; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in 0 1 0 1; do
  export BP_BF16_NT=$m
  d=$O/kt$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o kt -- python $R/tools/bench_bf16.py c5bf16 > $d.log 2>&1
  echo "== nontemporal W/delta stores $m"; grep -o '"ms_per_step": [0-9.]*' $d.log
  python - $d/kt_kernel_stats.csv <<'P'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'bf16' in r[0]: print('  ', r[0][:66].ljust(68), r[1], r[3][:8])
P
done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
