#!/bin/bash
# HBM traffic of the hot-path kernels from PMC counters, as MI355X_MICROARCH.md prescribes: separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), no trace domains next to --pmc.  Run on the GPU box:
#   bash tools/collect_traffic.sh [f32|bf16]      -> gpurun_out/traffic_<prec>/{fetch,write}/p_counter_collection.csv
# then `python tools/summarize_traffic.py f32` writes profiles/r01_traffic_<prec>.json (read by bench.py).
set -e
PREC=${1:-f32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/traffic_$PREC
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision $PREC > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision $PREC > $OUT.write.log 2>&1
ls $OUT/fetch $OUT/write
