#!/bin/bash
O=gpurun_out/bf1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_dp_native.py -m gpu -x -q -k "bf16" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for r in 1 2 3; do python tools/bench_bf16.py c5bf16 2>$O/err.txt | tail -1; done
