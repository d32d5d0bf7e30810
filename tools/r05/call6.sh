#!/bin/bash
# round 5, GPU call 6: the driver's bench command (new line pieces), --gpus 2 on one device (native + RCCL back to back),
# and the exchange kernel's momentum stream with nontemporal accesses (A/B)
O=gpurun_out/r05c6; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err ) 2>&1 | grep real
python - $O/bench_line.json <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = j["roofline"]
print("value", j["value"], "ms", j["ms_per_step"], "frac", r["frac"], "rocprof_frac", r["rocprof_frac"], "traffic", r["traffic"], r["traffic_stamp"].get("measured_in_this_run"))
print("mfma_util", {k: round(v["mfma_busy_frac"], 4) for k, v in r.get("mfma_util", {}).items() if isinstance(v, dict)}, r.get("live_counters_error"))
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["c1"]["value"], "c5", j["c5_bf16"].get("ms_per_step"), "dp_world1", j["dp_world1"].get("ms_per_step"), j["dp_world1"].get("vs_fused_step"))
PY
( time python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench2_line.json 2> $O/bench2.err ) 2>&1 | grep real
tail -c 900 $O/bench2_line.json; echo; tail -3 $O/bench2.err
w1() {
  local label=$1 lib=$2; shift 2
  ( export BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib "$@"
    for rep in 1 2; do
      timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2> $O/w1_$label.$rep.err | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('world-1 $label %.4f ms/step' % j['ms_per_step'])"
    done )
}
w1 dev libbp_hip_dev.so BP_X=1
w1 nt_delta libbp_hip_nt.so BP_X=1
w1 dev libbp_hip_dev.so BP_X=1
w1 nt_delta libbp_hip_nt.so BP_X=1
