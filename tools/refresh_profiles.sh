#!/bin/bash
# Regenerates the rocprofv3 summaries that profiles/ keeps (run on the GPU box; outputs under gpurun_out/).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof
mkdir -p $O
for P in f32 bf16 f16 f32x3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$P -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --precision $P > $O/bench_$P.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/l16 -o p -- python3 bench.py --workload l16_1024_tiled --steps 3 --warmup 1 > $O/bench_l16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_bf16 -o p -- python3 bench.py --mode train --precision bf16 --batch 64 --steps 3 --warmup 1 > $O/bench_train_bf16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_f32 -o p -- python3 bench.py --mode train --precision f32 --batch 16 --steps 3 --warmup 1 > $O/bench_train_f32.log 2>&1
tail -n 1 $O/bench_*.log
