#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dp_native.py tests/test_bpforward.py -m gpu -x -q -k "bf16 or bpforward" > $O/pytest_bf16.log 2>&1; echo "bf16 pytest rc=$?"; tail -2 $O/pytest_bf16.log
for i in 1 2; do timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_stg_$i.json 2>&1; echo "staged $(tail -1 $O/c5_stg_$i.json | cut -c60-160)"; done
timeout 120 python tools/bench_bf16.py c2bf16 > $O/c2_stg.json 2>&1; tail -1 $O/c2_stg.json
