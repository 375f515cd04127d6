#!/bin/bash
# ref_grid (the reference's published regime) on the GPU box: the bench line and rocprofv3 kernel stats of ViT-B/16 at batch 4 / batch 1.
# tools/ref_grid_profile.sh <tag>   -> gpurun_out/<tag>_ref_grid*.{json,csv}
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O/prof
python3 bench.py --workload ref_grid --steps 30 --warmup 3 > $O/${TAG}_ref_grid_f32.json 2> $O/${TAG}_ref_grid_f32.err
tail -c 600 $O/${TAG}_ref_grid_f32.json
for B in 4 1; do
  rm -rf $O/prof/rg_b$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/rg_b$B -o p -- python3 bench.py --workload ref_grid --grid-configs 0 --batch $B --steps 20 --warmup 2 --no-cpu-baseline > $O/prof/rg_b$B.log 2>&1
  cp $(find $O/prof/rg_b$B -name 'p_kernel_stats.csv' | head -1) $O/${TAG}_ref_grid_b16_batch${B}_kernel_stats.csv
  python3 tools/kstats.py $O/${TAG}_ref_grid_b16_batch${B}_kernel_stats.csv 30
done
