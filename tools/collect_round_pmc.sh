#!/bin/bash
# Counter evidence for the round's final kernels (run on the GPU box): the four passes of tools/collect_pmc.sh over the
# default inference bench (fp32 and bf16) and over one bf16 training step at batch 64.
#   python3 tools/summarize_pmc.py bench_f32 > profiles/rNN_pmc_bench_f32.json   (etc.) reduces them afterwards.
set -e
bash tools/collect_pmc.sh bench_f32 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision f32
bash tools/collect_pmc.sh bench_bf16 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision bf16
bash tools/collect_pmc.sh train_bf16 python3 bench.py --mode train --precision bf16 --batch 64 --steps 1 --warmup 1
