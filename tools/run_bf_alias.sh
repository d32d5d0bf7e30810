#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD; O=$R/gpurun_out/bfdma; rm -rf $O; mkdir -p $O
export BP_HIP_LIB=$R/dnn-for-speech-enhancement_amd/libbp_hip_dev.so      # (make -C dnn-for-speech-enhancement_amd/csrc dev: the build that reads BP_BF16_ROT_FWD)
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_dp_native.py tests/test_bptrain.py tests/test_bpforward.py -m gpu -x -q -k "bf16 or config5 or compute_dtype" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log
cd /tmp; export TMPDIR=/tmp
for m in 4 0 2 8 4 0; do
  export BP_BF16_ROT_FWD=$m
  d=$O/kt$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o kt -- python $R/tools/bench_bf16.py c5bf16 > $d.log 2>&1
  echo "== forward sharers $m tiles apart"; grep -o '"ms_per_step": [0-9.]*' $d.log
  python - $d/kt_kernel_stats.csv <<'P'
import csv,sys
for r in csv.reader(open(sys.argv[1])):
    if 'bf16<0' in r[0] or 'bf16<2' in r[0] or 'wgrad' in r[0]: print('  ', r[0][:66].ljust(68), r[1], r[3][:8])
P
done
find $O -name '*kernel_trace.csv' -delete; find $O -name '*.db' -delete
