#!/bin/bash
# GPU run 3 of round 3 (safe part): bf16 wgrad 128x64 tiles A/B, DP two-group exchange (tests + world-1 timing)
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dp_native.py tests/test_ref_bptrain.py tests/test_bptrain.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest.log
timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_tm2.json 2>&1; tail -1 $O/c5_tm2.json
BP_BF16_WGRAD_NO128=1 timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_tm1.json 2>&1; tail -1 $O/c5_tm1.json
timeout 200 python bench.py --gpus 1 --force-dp --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/bench1dp.json 2>$O/bench1dp.err; python - $O/bench1dp.json <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("dp world-1: %.4f ms/step %.0f frames/s" % (j["ms_per_step"], j["value"]))
PY
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench1.json 2>$O/bench1.err; python - $O/bench1.json <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("fused: %.4f ms/step %.0f frames/s" % (j["ms_per_step"], j["value"]), j["roofline"]["peak_measured"], "c5", j["c5_bf16"]["ms_per_step"])
PY
