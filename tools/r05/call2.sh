#!/bin/bash
# round 5, GPU call 2: (1) persistent double-buffered bf16 update kernel: parity + A/B + rocprof; (2) bp_dp_sync with the per-XCD
# release: DP tests + world-1 timing against the single-workgroup sync.
O=gpurun_out/r05c2; mkdir -p $O
DEV=$PWD/dnn-for-speech-enhancement_amd/libbp_hip_dev.so
echo "== bf16 parity with the persistent update kernel (dev lib, 512 workgroups)"
BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=512 timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or config5" > $O/pytest_bf16_persist.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_bf16_persist.log
echo "== configs[4] shape, bf16 step time"
for p in 0 256 512 768 0 512; do
  echo -n "persist=$p: "; BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=$p timeout 300 python tools/bench_bf16.py c5bf16 2>&1 | tail -1
done
echo "== rocprof kernel stats, persist=512"
cd /tmp; export TMPDIR=/tmp
BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=512 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_p512 -o p512 -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5bf16 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_p512 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/persist512_kernel_stats.csv && head -8 $f | cut -c1-160
echo "== data-parallel tests with the 16-workgroup release sync (product library)"
timeout 900 python -m pytest tests/test_dp_native.py -m gpu -x -q > $O/pytest_dp.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_dp.log
w1() {
  local label=$1 lib=$2; shift 2
  ( export BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib "$@"
    for rep in 1 2; do
      timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2> $O/w1_$label.$rep.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('world-1 $label %.4f ms/step' % j['ms_per_step'])"
    done )
}
w1 product_sync16 libbp_hip.so BP_X=1
w1 dev_sync16 libbp_hip_dev.so BP_X=1
w1 dev_sync1 libbp_hip_dev.so BP_DP_SYNC_WGS=1
w1 dev_sync8 libbp_hip_dev.so BP_DP_SYNC_WGS=8
timeout 300 python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused %.4f ms/step' % j['ms_per_step'])"
