#!/bin/bash
# round 5, GPU call 5: layer 1's gradient segment in pieces (tile-counter path): DP tests + world-1 timing against 1 / 2 pieces
O=gpurun_out/r05c5; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_native.py -m gpu -x -q > $O/pytest_dp.log 2>&1; echo "dp tests rc=$?"; tail -2 $O/pytest_dp.log
timeout 600 python -m pytest tests/test_ref_bptrain.py tests/test_bptrain.py -m gpu -x -q > $O/pytest_bptrain.log 2>&1; echo "bptrain tests rc=$?"; tail -2 $O/pytest_bptrain.log
w1() {
  local label=$1 lib=$2; shift 2
  ( export BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib "$@"
    for rep in 1 2; do
      timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2> $O/w1_$label.$rep.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('world-1 $label %.4f ms/step' % j['ms_per_step'])"
    done )
}
w1 product_4pieces libbp_hip.so BP_X=1
w1 dev_1piece libbp_hip_dev.so BP_DP_SLABS=1
w1 dev_2pieces libbp_hip_dev.so BP_DP_SLABS=2
w1 dev_3pieces libbp_hip_dev.so BP_DP_SLABS=3
w1 dev_4pieces_grid1_64 libbp_hip_dev.so BP_DP_GRID1=64
w1 dev_4pieces_grid192 libbp_hip_dev.so BP_DP_GRID=192
timeout 300 python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused %.4f ms/step' % j['ms_per_step'])"
echo "== timeline, 4 pieces"
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dp -o dp -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-dp --steps 60 --warmup 10 --no-cpu-baseline --no-extras --sustained-s 0 --prewarm-s 0 > /dev/null 2>&1 )
python tools/trace_timeline.py /tmp/prof_dp 70 > $O/dp_timeline.txt 2>&1; tail -45 $O/dp_timeline.txt
