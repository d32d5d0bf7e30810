#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 600 python -m pytest tests/test_bptrain.py tests/test_ref_bptrain.py tests/test_bpforward.py -m gpu -x -q > $O/pytest_bptrain.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_bptrain.log
timeout 600 python tools/bench_bptrain.py 4000 420 > $O/bptrain_pinned.json 2>$O/bptrain_pinned.err; cat $O/bptrain_pinned.json
