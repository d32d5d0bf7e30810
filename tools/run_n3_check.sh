#!/bin/bash
O=gpurun_out/n3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bptrain.py tests/test_ref_bptrain.py tests/test_bpforward.py tests/test_dp_native.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/bench_windows.py 2>$O/bw.err | tee $O/bench_windows.json
timeout 600 python tools/bench_bptrain.py 4000 420 2>$O/bt.err | tee $O/bptrain.json
