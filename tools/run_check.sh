#!/bin/bash
# the GPU test tier + smoke + the driver's bench command (what the round-end driver runs)
O=gpurun_out/check; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -c 1500 $O/bench_line.json; echo; tail -3 $O/bench.err
