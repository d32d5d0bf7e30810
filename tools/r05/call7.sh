#!/bin/bash
# round 5, GPU call 7: what the driver runs at round end (GPU test tier, smoke, bench) + the round-5 profile passes
O=gpurun_out/r05c7; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_r05.sh > $O/profile.log 2>&1; echo "profile rc=$?"
ls gpurun_out/prof_r05 | head -30
cat gpurun_out/prof_r05/mfma_util.txt | grep "bp_gemm\|wgrad"
