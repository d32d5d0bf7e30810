#!/bin/bash
# A/B of the merged dgrad+wgrad launches (BP_MERGE) on the C2 fp32 step, plus the parity tests under the switch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/merge
for rep in 1 2; do
  for m in 0 1 2; do
    echo "BP_MERGE=$m" >> gpurun_out/merge/ab.txt
    BP_MERGE=$m timeout 300 python tools/bench_bf16.py c2f32 >> gpurun_out/merge/ab.txt 2>&1
  done
done
BP_MERGE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py -m gpu -x -q > gpurun_out/merge/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/merge/ab.txt
cat gpurun_out/merge/ab.txt
