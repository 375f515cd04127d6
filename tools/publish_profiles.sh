#!/bin/bash
# Copies the rocprofv3 summaries that tools/refresh_profiles.sh + tools/collect_round_pmc.sh left under gpurun_out/
# into profiles/ under the round's prefix:   bash tools/publish_profiles.sh r03
set -e
R=${1:?round prefix, e.g. r03}
O=gpurun_out/prof
declare -A NAME=([f32]=bench_f32 [bf16]=bench_bf16 [f16]=bench_f16 [f32x3]=bench_f32x3 [l16]=bench_l16_1024_tiled_f16 [train_bf16]=bench_train_bf16_B64 [train_f32]=bench_train_f32_B16)
for k in "${!NAME[@]}"; do
  [ -f $O/$k/p_kernel_stats.csv ] || continue
  cp $O/$k/p_kernel_stats.csv profiles/${R}_${NAME[$k]}_kernel_stats.csv
  grep '^{"metric' $O/bench_$k.log > profiles/${R}_${NAME[$k]}_under_rocprof.json
done
for t in bench_f32 bench_bf16 train_bf16; do
  [ -d gpurun_out/pmc_$t ] || continue
  python3 tools/summarize_pmc.py $t > profiles/${R}_pmc_$t.json
done
python3 tools/summarize_traffic.py f32 $R
python3 tools/summarize_traffic.py bf16 $R
