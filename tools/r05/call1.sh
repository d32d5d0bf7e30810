#!/bin/bash
# round 5, GPU call 1: the refactored library through the whole GPU tier + the driver's bench command, then A/B of the
# data-parallel world-1 path (tile counter release vs relaxed; exchange grids) with the development builds.
O=gpurun_out/r05c1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench_line.json; echo
w1() {  # label, lib, env...
  local label=$1 lib=$2; shift 2
  ( export BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib "$@"
    for rep in 1 2; do
      timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 > $O/w1_$label.$rep.json 2> $O/w1_$label.$rep.err
      python - $O/w1_$label.$rep.json "$label" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("world-1", sys.argv[2], "%.4f ms/step" % j["ms_per_step"])
except Exception as e:
    print("world-1", sys.argv[2], "failed", e)
PY
    done )
}
w1 release libbp_hip_dev.so BP_X=1
w1 relaxed libbp_hip_relaxed.so BP_X=1
for g1 in 48 64 96; do w1 grid1_$g1 libbp_hip_dev.so BP_DP_GRID1=$g1; done
w1 grid96 libbp_hip_dev.so BP_DP_GRID=96
w1 events libbp_hip_dev.so BP_DP_NO_COUNTERS=1
# fused step for the same box
timeout 300 python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused %.4f ms/step' % j['ms_per_step'])"
