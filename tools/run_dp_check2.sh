#!/bin/bash
O=gpurun_out/dp7; mkdir -p $O
for v in inline kernel; do
  if [ $v = kernel ]; then export BP_DP_NO_INLINE_WAIT=1; else unset BP_DP_NO_INLINE_WAIT; fi
  for rep in 1 2; do
    timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 > $O/w1_${v}_$rep.json 2> $O/w1_${v}_$rep.err
    python - $O/w1_${v}_$rep.json $v <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("wait:", sys.argv[2], "world-1 exchange path: %.4f ms/step" % j["ms_per_step"])
except Exception as e:
    print("wait:", sys.argv[2], "failed", e)
PY
  done
done
unset BP_DP_NO_INLINE_WAIT
timeout 900 python -m pytest tests/test_dp_native.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused single-device step: %.4f ms' % j['ms_per_step'])"
