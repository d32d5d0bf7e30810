#!/bin/bash
# GPU run 2 of round 3: bf16 LDS-DMA wgrad validation + timing, copy probe, output-layer split A/B, DP world-1 timeline
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_dp_native.py -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; echo "bf16 pytest rc=$?" | tee $O/rc.txt; tail -3 $O/pytest_bf16.log
python tools/bench_bf16.py c5bf16 > $O/c5_new.json 2>&1; tail -1 $O/c5_new.json
BP_BF16_NO_DMA=1 python tools/bench_bf16.py c5bf16 > $O/c5_old.json 2>&1; tail -1 $O/c5_old.json
python tools/bench_bf16.py c2bf16 > $O/c2_new.json 2>&1; tail -1 $O/c2_new.json
tools/copy_probe.bin | tee $O/copy_probe.json
for s in 4 8 16; do BP_OUT_SPLITS=$s python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_out$s.json 2>$O/bench_out$s.err; python - $O/bench_out$s.json $s <<'PY'
import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k=j["roofline"]["kernels_in_step_ms"]
print("out_splits", sys.argv[2], "ms/step %.4f" % j["ms_per_step"], "fwd_out %.2f us dgrad_out %.2f us" % (1e3*k["fwd_out"], 1e3*k["dgrad_out"]))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/dp1 -o dp1 -- python $GRAFT_REPO_ROOT/bench.py --force-dp --steps 30 --warmup 5 --prewarm-s 0.3 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/dp1.out 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_timeline.py $O/dp1 120 > $O/dp1_timeline.txt 2>&1; find $O/dp1 -name "*.csv" -size +20M -delete; tail -70 $O/dp1_timeline.txt
