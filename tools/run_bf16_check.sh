#!/bin/bash
O=gpurun_out/bf2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_dp_native.py -m gpu -x -q -k "bf16" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in split nosplit split nosplit; do
  if [ $v = nosplit ]; then export BP_BF16_NO_SPLITK=1; else unset BP_BF16_NO_SPLITK; fi
  echo $v; python tools/bench_bf16.py c5bf16 2>$O/err.txt | tail -1
done
