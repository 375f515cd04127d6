#!/bin/bash
# A/B of the bf16 attention kernels at the training geometry (B = 64, Np = 1024, A = 12): per-kernel durations from
# rocprofv3 --kernel-trace --stats of the forward + backward probe, serial round-2 forward loop against the pipelined one.
#   bash tools/attn_ab.sh [drop]      (drop = 0.1 --words by default)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
DROP=${1:-0.1}
ARGS="tools/op_probe.py attn_bwd --B 64 --Np 1024 --A 12 --iters 5 --drop $DROP"
[ "$DROP" != "0" ] && ARGS="$ARGS --words"
for v in pipe serial; do
  OUT=gpurun_out/attn_ab_${v}_$DROP
  rm -rf $OUT
  if [ $v = serial ]; then export VITSEG_ATTN_SERIAL=1; else unset VITSEG_ATTN_SERIAL; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $ARGS > $OUT.log 2>&1
  echo "== $v (dropout $DROP)"
  python3 tools/kstats.py $(find $OUT -name "*kernel_stats.csv" | head -1) 8
done
