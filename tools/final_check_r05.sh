#!/bin/bash
O=gpurun_out/r05final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; grep -E "passed|failed" $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_r05.sh > $O/profile.log 2>&1; echo "profile rc=$?"
cat gpurun_out/prof_r05/stamp.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json
