for M in 32768 32800 32896 65536; do for K in 768 3072; do echo "f32 M=$M N=768 K=$K: $(python3 tools/op_probe.py linear --M $M --N 768 --K $K --epi 0 --iters 20 | tail -1)"; done; done
for N in 1536 3072; do echo "f32 M=32768 N=$N K=3072: $(python3 tools/op_probe.py linear --M 32768 --N $N --K 3072 --epi 0 --iters 10 | tail -1)"; done
