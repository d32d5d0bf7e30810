#!/bin/bash
# GPU run 4 of round 3: bf16 C5 kernel-level stats, DP world-1 timeline after the two-group change, end-to-end bptrain rate
# after the reader change (bounded memory: 2000 x 420 frames = 2 x 0.9 GB of Pfiles)
O=gpurun_out/r3d; mkdir -p $O
free -g | tee $O/mem.txt; df -h /tmp | tee -a $O/mem.txt; nproc | tee -a $O/mem.txt
timeout 400 python tools/bench_bptrain.py 2000 420 > $O/bptrain.json 2>$O/bptrain.err; cat $O/bptrain.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5bf16 > $GRAFT_REPO_ROOT/$O/c5.out 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/dp1 -o dp1 -- python $GRAFT_REPO_ROOT/bench.py --force-dp --steps 30 --warmup 5 --prewarm-s 0.3 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/dp1.out 2>&1
cd $GRAFT_REPO_ROOT; python tools/trace_timeline.py $O/dp1 90 > $O/dp1_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/r3d/c5/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
tail -45 $O/dp1_timeline.txt
