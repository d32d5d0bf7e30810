#!/bin/bash
# One command for the first lease with more than one MI355X (nothing here has ever run across devices; SCALE was skipped
# in rounds 1-2).  Writes everything under gpurun_out/multi_gpu/.  Order = cheapest evidence first.
#   1. the fabric and RCCL: per-link / all-link copy rates, kernel peer reads+writes, RCCL reduce-scatter+all-gather of the
#      58.84 MB C4 message (tools/xgmi_probe.hip)
#   2. correctness ACROSS devices: the DP tests spread their ranks over all visible devices automatically
#      (tests/dp_worker.py), plus the test that requires >= 2 devices
#   3. the scaling curve of the driver's own command, native exchange and RCCL transport side by side
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/multi_gpu; mkdir -p $OUT
N=$(python3 -c "import dnnse_amd; print(dnnse_amd.device_count())")
echo "visible devices: $N" | tee $OUT/summary.txt
[ -x tools/xgmi_probe.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/xgmi_probe.bin tools/xgmi_probe.hip -L/opt/rocm/lib -lrccl
timeout 300 tools/xgmi_probe.bin > $OUT/xgmi_probe.json 2> $OUT/xgmi_probe.err; tail -1 $OUT/xgmi_probe.json | tee -a $OUT/summary.txt
timeout 1500 python3 -m pytest tests/test_dp_native.py tests/test_ref_bptrain.py -m gpu -x -q -s > $OUT/pytest_dp.log 2>&1; tail -3 $OUT/pytest_dp.log | tee -a $OUT/summary.txt
for n in 1 2 4 8; do
  [ $n -le $N ] || continue
  for ex in native rccl; do
    [ $n -eq 1 ] && [ $ex = rccl ] && continue
    timeout 600 python3 bench.py --gpus $n --steps 200 --warmup 20 --exchange $ex --no-cpu-baseline --no-extras > $OUT/bench_n${n}_${ex}.json 2> $OUT/bench_n${n}_${ex}.err
    python3 - $OUT/bench_n${n}_${ex}.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("n=%d %-6s %.0f frames/s  %.4f ms/step  distinct devices %s" % (j["n_gpus"], j["config"]["exchange"][:6], j["value"], j["ms_per_step"], j.get("distinct_devices")))
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  done
done
