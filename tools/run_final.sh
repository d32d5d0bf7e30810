#!/bin/bash
# what the driver runs at round end: the GPU test tier, smoke(), the bench at N=1 (its exact command), plus N=2 (ranks sharing the device)
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json; echo
python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench2_line.json 2> $O/bench2.err; tail -c 300 $O/bench2_line.json; echo
timeout 600 python tools/bench_bptrain.py 4000 420 > $O/bptrain.json 2>$O/bptrain.err; cat $O/bptrain.json
python tools/bench_windows.py > $O/bench_windows.json 2> $O/bench_windows.err; cat $O/bench_windows.json
