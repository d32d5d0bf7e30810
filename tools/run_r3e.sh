#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dp_native.py -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; echo "bf16 pytest rc=$?"; tail -4 $O/pytest_bf16.log
for i in 1 2; do timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_w8_$i.json 2>&1; echo "w8 $(tail -1 $O/c5_w8_$i.json | cut -c60-160)"; done
BP_BF16_NO_W8=1 timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_now8.json 2>&1; echo "4 waves $(tail -1 $O/c5_now8.json | cut -c60-160)"
