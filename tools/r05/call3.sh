#!/bin/bash
# round 5, GPU call 3: persistent bf16 update kernel with FOUR update waves (two workgroups per CU by construction); the
# data-parallel gradient store with system-scope write-through stores (no release, single-wave sync).
O=gpurun_out/r05c3; mkdir -p $O
DEV=$PWD/dnn-for-speech-enhancement_amd/libbp_hip_dev.so
echo "== bf16 parity, persistent kernel, 4 update waves, 512 workgroups"
BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=512 timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or config5" > $O/pytest_bf16_persist.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_bf16_persist.log
echo "== configs[4] shape, bf16 step time"
for p in 0 512 768 256 0 512; do
  echo -n "persist=$p waves=4: "; BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=$p timeout 300 python tools/bench_bf16.py c5bf16 2>&1 | tail -1 | cut -c60-140
done
echo -n "persist=512 waves=2: "; BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=512 BP_BF16_UPD_WAVES=2 timeout 300 python tools/bench_bf16.py c5bf16 2>&1 | tail -1 | cut -c60-140
echo "== rocprof kernel stats, persist=512"
( cd /tmp; export TMPDIR=/tmp; BP_HIP_LIB=$DEV BP_BF16_UPD_PERSIST=512 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p512 -o p512 -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5bf16 > /dev/null 2>&1 )
f=$(find /tmp/prof_p512 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/persist512_kernel_stats.csv && head -6 $f | cut -c1-150
echo "== data-parallel tests, write-through gradient stores (product library)"
timeout 900 python -m pytest tests/test_dp_native.py -m gpu -x -q > $O/pytest_dp.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_dp.log
w1() {
  local label=$1 lib=$2; shift 2
  ( export BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib "$@"
    for rep in 1 2; do
      timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2> $O/w1_$label.$rep.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('world-1 $label %.4f ms/step' % j['ms_per_step'])"
    done )
}
w1 product libbp_hip.so BP_X=1
w1 events libbp_hip_dev.so BP_DP_NO_COUNTERS=1
timeout 300 python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('fused %.4f ms/step' % j['ms_per_step'])"
echo "== the new bench line pieces (live counter passes, c1 cpu baseline)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; python - $O/bench_line.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = j["roofline"]
print("value", j["value"], "ms", j["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_stamp"].get("measured_in_this_run"))
print("mfma_util", r.get("mfma_util"), r.get("live_counters_error"))
print("hidden", r["hidden_fwd_2048x2048"])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("c1"))
print("c5", j["c5_bf16"].get("ms_per_step"), "dp_world1", j["dp_world1"])
PY
