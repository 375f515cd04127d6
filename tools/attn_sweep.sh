python3 tools/op_probe.py attention --iters 20 | tail -1
python3 tools/op_probe.py attention --bf16 --iters 20 | tail -1
python3 -m pytest tests -x -q -m gpu -k "attention or attn or backward" 2>&1 | tail -3
python3 bench.py --no-cpu-baseline | tail -1
python3 bench.py --no-cpu-baseline --precision bf16 | tail -1
python3 bench.py --no-cpu-baseline --mode train --precision bf16 --batch 64 | tail -1
