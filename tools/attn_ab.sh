#!/bin/bash
# Per-kernel durations of the bf16 attention kernels at the training geometry (B = 64, Np = 1024, A = 12) from
# rocprofv3 --kernel-trace --stats of the forward + backward probe.   bash tools/attn_ab.sh [drop] [tag]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
DROP=${1:-0.1}
TAG=${2:-cur}
ARGS="tools/op_probe.py attn_bwd --B 64 --Np 1024 --A 12 --iters 5 --drop $DROP"
[ "$DROP" != "0" ] && ARGS="$ARGS --words"
OUT=gpurun_out/attn_ab_${TAG}_$DROP
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ARGS > $OUT.log 2>&1
echo "== $TAG (dropout $DROP)"
python3 tools/kstats.py $(find $OUT -name "*kernel_stats.csv" | head -1) 8
