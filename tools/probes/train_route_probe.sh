#!/bin/bash
# fp32 training step of ViT-B/16 at 224x224 by batch: the small-batch route against the large-batch kernels (VITSEG_NO_SMALL=1).
# bash tools/probes/train_route_probe.sh > gpurun_out/<tag>.txt
set -e
for B in 4 8 16 32 64; do
  for NS in 0 1; do
    echo -n "no_small=$NS "; VITSEG_NO_SMALL=$NS python tools/train_small_probe.py fp32 10 $B 2>/dev/null | tail -n 1
  done
done
