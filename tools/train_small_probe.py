#!/usr/bin/env python3
"""The reference's TRAINING regime (batch 4 x 224x224, model/CE/trainCurrentViTmodel.py:57): steps of ViT-B/16, 17 classes, for
rocprofv3 --kernel-trace --stats.  python tools/train_small_probe.py [fp32|bf16] [steps] [batch]   (VITSEG_NO_SMALL=1: the large-batch kernels)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import synth  # noqa: E402
from visiontransformer_amd.config import ViTSegConfig  # noqa: E402
from visiontransformer_amd.model import ViTSegmentationModel  # noqa: E402
from visiontransformer_amd.optim import FusedAdam  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda:0"
cfg = ViTSegConfig(17, 16, 768, 12, 12, image_size=224)
m = ViTSegmentationModel(17, 16, 768, 12, 12, image_size=224, precision=prec, dropout=0.1, device=dev).train()
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=1).items()})
x = torch.from_numpy(synth.make_images(cfg, B, seed=0)).to(dev)
y = torch.from_numpy(synth.make_targets(cfg, B, seed=0, size=224)).to(dev)
opt = FusedAdam(m.parameters(), lr=1e-5)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.ce_loss(x, y, grad_scale=1.0)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    loss = step()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(steps):
    loss = step()
t1.record()
torch.cuda.synchronize()
print("batch", B, "loss", float(loss), "ms/step %.3f" % (t0.elapsed_time(t1) / steps))
