#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
for v in base f32 f32all d32 fd32 base; do
  if [ $v = base ]; then unset BP_HIP_LIB; else export BP_HIP_LIB=$PWD/ab_libs/libbp_$v.so; fi
  timeout 200 python bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline > $O/b_$v.json 2>$O/b_$v.err; python - $O/b_$v.json $v <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k=j["roofline"]["kernels_in_step_ms"]
print(sys.argv[2], "ms/step %.4f" % j["ms_per_step"], {a: round(1e3*b,1) for a,b in k.items()})
PY
done
