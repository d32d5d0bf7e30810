# A/B of prebuilt library variants under ab_libs/ (timing only): bash tools/ab_libs.sh v1 v2 ...
L=dnn-for-speech-enhancement_amd/libbp_hip.so
cp $L /tmp/keep.so
for v in "$@"; do
  cp ab_libs/libbp_hip_$v.so $L
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['value']), round(r['ms_per_step'],4), {k: round(v*1e3,1) for k,v in r['roofline']['kernel_ms'].items()})"
done
cp /tmp/keep.so $L
