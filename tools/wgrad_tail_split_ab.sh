R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/split_ab; mkdir -p $O; rm -f $O/ab3.txt
export BP_HIP_LIB=$R/dnn-for-speech-enhancement_amd/libbp_hip_dev.so
B="--steps 400 --warmup 40 --no-cpu-baseline --no-extras --prewarm-s 1.0 --sustained-s 0"
for s in 0 8 256 576 1024 1536 2048 3648 0 576 1024 1536; do
  BP_WGRAD_SPLIT=$s python $R/bench.py $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $s', d['ms_per_step'])" >> $O/ab3.txt
done
cat $O/ab3.txt
