#!/bin/bash
# copies the summaries of tools/profile_r06.sh (run through gpurun: `bash tools/profile_r06.sh`) (gpurun_out/prof_r06, scratch) into profiles/ (tracked)
S=gpurun_out/prof_r06; D=profiles
cp $(find $S/kt -name "*kernel_stats.csv" | head -1) $D/r06_bench_kernel_stats.csv
cp $S/bench_under_rocprof.json $D/r06_bench_under_rocprof.json
cp $S/pmc.json $D/r06_pmc_hbm_traffic.json
cp $S/sq.json $D/r06_pmc_sq_counters.json; cp $S/sq.txt $D/r06_pmc_sq_counters.txt
cp $S/mfma_util.json $D/r06_mfma_util.json
cp $(find $S/c5 -name "*kernel_stats.csv" | head -1) $D/r06_bf16_c5_kernel_stats.csv
tail -1 $S/c5.json > $D/r06_bf16_c5_line.json
cp $S/c5_pmc.json $D/r06_c5_pmc_hbm_traffic.json
cp $S/c5_sq.json $D/r06_c5_pmc_sq_counters.json; cp $S/c5_sq.txt $D/r06_c5_pmc_sq_counters.txt
cp $S/c5_mfma_util.json $D/r06_c5_mfma_util.json
echo "stamp $(cat $S/stamp.txt); current sources $(cat dnn-for-speech-enhancement_amd/csrc/*.h dnn-for-speech-enhancement_amd/csrc/*.hip | sha256sum | cut -c1-16)"
