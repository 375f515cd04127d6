#!/bin/bash
# Where does the 16-bit form of the small-batch route stop paying?  ViT-B/16 at 197 and 785 tokens, bf16, by batch:
# the large-batch kernels (VITSEG_NO_SMALL=1) against the route forced on (VITSEG_SMALL_MAX_ROWS=16384).
# bash tools/probes/h16_route_probe.sh > gpurun_out/<tag>.txt   (profiles/r05_h16_route_probe*.txt)
set -e
for B in 1 2 4 8 16 32; do
  for MODE in "VITSEG_NO_SMALL=1" "VITSEG_SMALL_MAX_ROWS=16384"; do
    env $MODE python bench.py --workload ref_grid --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline --grid-configs 0,4 --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for g in d['grid']:
    k=[k for k in g if k.startswith('batch')][0]
    print('B=$B $MODE', g['config'], g['tokens'], k, g[k]['ms_eager'])
"
  done
done
