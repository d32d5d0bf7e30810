#!/bin/bash
# One command for the first lease with more than one MI355X (nothing here has ever run across devices; SCALE was skipped
# in rounds 1-2).  Writes everything under gpurun_out/multi_gpu/.  Order = cheapest evidence first.
#   1. the fabric and RCCL: per-link / all-link copy rates, kernel peer reads+writes, RCCL reduce-scatter+all-gather of the
#      58.84 MB C4 message (tools/xgmi_probe.hip)
#   2. correctness ACROSS devices: the DP tests spread their ranks over all visible devices automatically
#      (tests/dp_worker.py), plus the test that requires >= 2 devices
#   3. the scaling curve of the driver's own command (each N > 1 line times the native exchange and the RCCL transport side by side)
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
  # (N > 1: one run times BOTH transports back to back and takes the faster as the line's value: `exchange: {native, rccl, chosen}`)
  timeout 900 python3 bench.py --gpus $n --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $OUT/bench_n${n}.json 2> $OUT/bench_n${n}.err
  python3 - $OUT/bench_n${n}.json <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    ex = j.get("exchange") or {}
    per = "  ".join("%s %s" % (k, ("%.4f ms" % v["ms_per_step"]) if "ms_per_step" in v else "FAILED (%s)" % v.get("error", "?")[:60]) for k, v in ex.items() if isinstance(v, dict))
    print("n=%d %.0f frames/s  %.4f ms/step  chosen %s  [%s]  distinct devices %s" % (j["n_gpus"], j["value"], j["ms_per_step"], ex.get("chosen"), per, j.get("distinct_devices")))
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
done
