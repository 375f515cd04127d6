#!/usr/bin/env python3
"""Decoder-tail kernel alone: logits only / mask only / both, on near-tied (random-init-like) and well-separated logits."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib
L = _lib.lib(); dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
B, C, g, S = 32, int(os.environ.get("C", "2")), 32, 512
for name, scale in (("near-tied", 0.05), ("separated", 3.0)):
    z = (torch.randn(B, C, g, g, device=dev) * scale).contiguous()
    lg = torch.empty(B, C, S, S, device=dev); mk = torch.empty(B, S, S, dtype=torch.uint8, device=dev)
    for what, a, b, bytes_ in (("logits", lg.data_ptr(), None, lg.numel() * 4), ("mask", None, mk.data_ptr(), mk.numel()),
                               ("both", lg.data_ptr(), mk.data_ptr(), lg.numel() * 4 + mk.numel())):
        for _ in range(5):
            _lib.check(L.vitseg_op_upsample_argmax(z.data_ptr(), a, b, B, C, g, S, st))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            _lib.check(L.vitseg_op_upsample_argmax(z.data_ptr(), a, b, B, C, g, S, st))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"C={C} {name:10s} {what:6s}: {dt * 1e6:7.1f} us  {bytes_ / dt / 1e9:7.0f} GB/s")
