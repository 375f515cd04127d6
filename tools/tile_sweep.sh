# 16-bit GEMM tile variants on the model's shapes (M = 32800): VITSEG_BF16_TILES = small | large | xl | 4w
for t in small large xl 4w; do for shp in "3072 768 1" "2304 768 0" "768 3072 2" "768 768 2"; do set -- $shp; echo "$t N=$1 K=$2 epi=$3: $(VITSEG_BF16_TILES=$t python3 tools/op_probe.py linear --bf16 --M 32800 --N $1 --K $2 --epi $3 --iters 30 | tail -1)"; done; done
