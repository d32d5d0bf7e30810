#!/bin/bash
# round 5, GPU call 4: bf16 update launches on a second stream beside the dgrad GEMMs (BP_BF16_OVERLAP, dev build): parity, A/B, timeline
O=gpurun_out/r05c4; mkdir -p $O
DEV=$PWD/dnn-for-speech-enhancement_amd/libbp_hip_dev.so
echo "== bf16 parity with the overlapped schedule"
BP_HIP_LIB=$DEV BP_BF16_OVERLAP=1 timeout 900 python -m pytest tests -m gpu -x -q -k "bf16 or config5" > $O/pytest_overlap.log 2>&1; echo "rc=$?"; tail -2 $O/pytest_overlap.log
echo "== configs[4] shape, bf16 step time"
for v in 0 1 0 1 s 1; do
  ( [ $v = 1 ] && export BP_BF16_OVERLAP=1; [ $v = s ] && export BP_BF16_OVERLAP=1 BP_BF16_OVERLAP_SAMEPRIO=1
    echo -n "overlap=$v: "; BP_HIP_LIB=$DEV timeout 300 python tools/bench_bf16.py c5bf16 2>&1 | tail -1 | cut -c60-140 )
done
echo "== rocprof timeline, overlap=1"
( cd /tmp; export TMPDIR=/tmp; BP_HIP_LIB=$DEV BP_BF16_OVERLAP=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ov -o ov -- python $GRAFT_REPO_ROOT/tools/bench_bf16.py c5bf16 > /dev/null 2>&1 )
f=$(find /tmp/prof_ov -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/overlap_kernel_stats.csv && head -9 $f | cut -c1-150
python tools/trace_timeline.py /tmp/prof_ov 90 > $O/overlap_timeline.txt 2>&1; tail -64 $O/overlap_timeline.txt
