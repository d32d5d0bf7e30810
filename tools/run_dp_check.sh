#!/bin/bash
# data-parallel exchange path: the N-process tests (ranks share the one device) + the world-1 step time, with the in-kernel
# tile counters (default) and with the event + kernel-boundary hand-off (BP_DP_NO_COUNTERS=1)
O=gpurun_out/dp4; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_native.py tests/test_ref_bptrain.py -m gpu -x -q > $O/pytest_dp.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_dp.log
for v in counters events; do
  if [ $v = events ]; then export BP_DP_NO_COUNTERS=1; else unset BP_DP_NO_COUNTERS; fi
  for rep in 1 2; do
    python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 > $O/w1_${v}_$rep.json 2> $O/w1_${v}_$rep.err
    python - $O/w1_${v}_$rep.json $v <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "world-1 exchange path: %.4f ms/step" % j["ms_per_step"])
PY
  done
done
unset BP_DP_NO_COUNTERS
python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused single-device step: %.4f ms' % j['ms_per_step'])"
python bench.py --gpus 2 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --sustained-s 0 2>$O/w2.err | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2 ranks on one device: %.4f ms/step' % j['ms_per_step'])"
