#!/bin/bash
O=gpurun_out/dp8; mkdir -p $O
export BP_DP_NO_INLINE_WAIT=1
for g in 64 128 192 256 384; do
  export BP_DP_GRID=$g
  timeout 300 python bench.py --gpus 1 --force-dp --steps 400 --warmup 40 --no-cpu-baseline --no-extras --sustained-s 0 > $O/w1_g$g.json 2> $O/w1_g$g.err
  python - $O/w1_g$g.json $g <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("update grid", sys.argv[2], "world-1 exchange path: %.4f ms/step" % j["ms_per_step"])
except Exception as e:
    print("grid", sys.argv[2], "failed", e)
PY
done
