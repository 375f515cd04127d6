set -e
for B in 2 8 16 32; do
  for NS in 0 1; do
    VITSEG_NO_SMALL=$NS python bench.py --workload ref_grid --precision bf16 --steps 20 --warmup 3 --no-cpu-baseline --grid-configs 0,4 --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for g in d['grid']:
    k=[k for k in g if k.startswith('batch')][0]
    print('B=$B no_small=$NS', g['config'], g['tokens'], k, g[k]['ms_eager'])
"
  done
done
