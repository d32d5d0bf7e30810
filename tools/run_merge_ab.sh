#!/bin/bash
# A/B of a development switch on the C2 fp32 step (alternating, same box): SW=<env var> VALS="0 1 ..."
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/ab.txt
for rep in 1 2 3; do
  for m in ${VALS:-0 1}; do
    if [ $m != 0 ]; then export ${SW:-BP_WG5}=$m; else unset ${SW:-BP_WG5}; fi
    echo "${SW:-BP_WG5}=$m $(timeout 300 python tools/bench_bf16.py c2f32 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')" >> gpurun_out/ab/ab.txt
  done
done
cat gpurun_out/ab/ab.txt
[ -n "$PYTEST" ] && { export ${SW}=$PYTEST; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "train_matches or full_size or golden" > gpurun_out/ab/pytest.log 2>&1; echo "pytest ($SW=$PYTEST) rc=$?"; grep -E "passed|failed" gpurun_out/ab/pytest.log; }
