R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_bf16; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/bench_bf16.py c2bf16 > $O/out.txt 2> $O/err.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
