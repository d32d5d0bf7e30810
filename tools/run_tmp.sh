R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icp; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export BP_HIP_LIB=$R/dnn-for-speech-enhancement_amd/libbp_hip_dev.so
for m in none 1 64; do
  if [ $m = none ]; then unset BP_ICACHE_PROBE; else export BP_ICACHE_PROBE=$m; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$m -o kt -- python $R/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extras --prewarm-s 0.5 --sustained-s 0 > $O/b$m.json 2> $O/kt$m.err
  echo "== probe $m: $(tail -1 $O/b$m.json | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")" >> $O/r.txt
  python - $O/kt$m >> $O/r.txt <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if n.startswith('void bp_') or 'icache' in n or 'bp_out' in n:
        if 'fill' in n or 'dp' in n: continue
        print('   %8.2f us  %6s calls  %s' % (float(r['AverageNs'])/1e3, r['Calls'], n[:90]))
PY
  find $O/kt$m -name "*kernel_trace.csv" -delete; find $O/kt$m -name "*.db" -delete
done
cat $O/r.txt
