#!/usr/bin/env python3
"""Serving-size latency of the forward: eager launch sequence vs hipGraph replay (predict_mask_graphed)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import synth  # noqa: E402
from visiontransformer_amd.config import ViTSegConfig  # noqa: E402
from visiontransformer_amd.model import ViTSegmentationModel  # noqa: E402

dev = "cuda:0"
for S in (224, 512):
    cfg = ViTSegConfig(17, 16, 768, 12, 12, image_size=S)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=1).items()}
    for prec in ("fp32", "fp32x3", "bf16"):
        m = ViTSegmentationModel(17, 16, 768, 12, 12, image_size=S, precision=prec, device=dev).eval()
        m.load_state_dict(sd)
        for B in (1, 4, 8):
            x = torch.rand(B, 3, S, S, device=dev)
            res = {}
            for name, fn in (("eager", m.predict_mask), ("graph", m.predict_mask_graphed)):
                for _ in range(5):
                    fn(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(50):
                    fn(x)
                torch.cuda.synchronize()
                res[name] = (time.perf_counter() - t0) / 50 * 1e3
            print(f"ViT-B/16 @{S} {prec:6s} batch {B}: eager {res['eager']:.3f} ms, graph {res['graph']:.3f} ms")
