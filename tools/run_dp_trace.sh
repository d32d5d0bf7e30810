#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/dp6; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --force-dp --steps 60 --warmup 20 --no-cpu-baseline --no-extras --sustained-s 0 --prewarm-s 0 > $O/bench.json 2> $O/err.txt
python $GRAFT_REPO_ROOT/tools/trace_timeline.py $O/kt 90 > $O/timeline.txt; find $O -name "*.csv" -delete; find $O -name "*.db" -delete
head -75 $O/timeline.txt
