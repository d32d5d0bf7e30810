#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dp_native.py -m gpu -x -q -k "bf16" > $O/pytest_bf16.log 2>&1; echo "bf16 pytest rc=$?"; tail -4 $O/pytest_bf16.log
for i in 1 2; do timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_sk_$i.json 2>&1; echo "splitk $(tail -1 $O/c5_sk_$i.json | cut -c60-160)"; done
BP_BF16_NO_SPLITK=1 timeout 120 python tools/bench_bf16.py c5bf16 > $O/c5_nosk.json 2>&1; echo "old gemm $(tail -1 $O/c5_nosk.json | cut -c60-160)"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt_sk -o kt -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5bf16 > $GRAFT_REPO_ROOT/$O/c5_sk_prof.out 2>&1
python - $GRAFT_REPO_ROOT/$O/kt_sk <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+"/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]: print("  ", r["Name"][:70], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
find $GRAFT_REPO_ROOT/$O -name "*kernel_trace.csv" -delete; find $GRAFT_REPO_ROOT/$O -name "*.db" -delete
