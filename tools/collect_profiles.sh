#!/bin/bash
# Copy the summaries tools/profile_round.sh left under gpurun_out/<tag>/ into profiles/ (tracked).  usage: bash tools/collect_profiles.sh r04
TAG=${1:-r04}; S=gpurun_out/$TAG; D=profiles
for c in f32 bf16 bf16_b32 bf16_512 bf16_1024; do
  cp $S/stats_$c/k_kernel_stats.csv $D/${TAG}_bench_${c}_kernel_stats.csv
  cp $S/stats_$c.json $D/${TAG}_bench_${c}_under_rocprof.json
done
cp $S/stats_dead/k_kernel_stats.csv $D/${TAG}_dead_scan_kernel_stats.csv
cp $S/dead_scan.json $D/${TAG}_dead_scan.json
cp $S/hbm_traffic_f32.json $D/${TAG}_hbm_traffic_f32.json
cp $S/hbm_traffic_bf16.json $D/${TAG}_hbm_traffic_bf16.json
[ -f $S/pmc_wino4.txt ] && cp $S/pmc_wino4.txt $D/${TAG}_pmc_wino4.txt
[ -f $S/pmc_mfma_bf16.txt ] && cp $S/pmc_mfma_bf16.txt $D/${TAG}_pmc_mfma_bf16.txt
[ -f gpurun_out/parity_measured.json ] && cp gpurun_out/parity_measured.json $D/${TAG}_parity_measured.json
[ -f gpurun_out/classifier_flips.json ] && cp gpurun_out/classifier_flips.json $D/${TAG}_classifier_flips.json
ls -la $D | grep ${TAG}_
