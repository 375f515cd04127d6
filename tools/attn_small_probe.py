#!/usr/bin/env python3
"""attention_small.hip (key-split blocks of 32 queries) against attention_f32.hip (128-query blocks, serial key loop) per
sequence length and batch: us per launch, 100 back-to-back launches.  python tools/attn_small_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

dev, L = "cuda:0", _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for Np, A in ((196, 12), (784, 12), (1024, 12), (3136, 12)):
    for B in (1, 2, 4, 8, 16):
        if B * (Np + 1) > 16500:
            continue
        rows, D = B * Np + B, 64 * A
        qkv = torch.randn(rows, 3 * D, device=dev)
        ctx = torch.empty(rows, D, device=dev)
        ts = timed(lambda: _lib.check(L.vitseg_op_attention_f32_small(qkv.data_ptr(), ctx.data_ptr(), B, Np, A, st())))
        tb = timed(lambda: _lib.check(L.vitseg_op_attention_f32(qkv.data_ptr(), ctx.data_ptr(), B, Np, A, st())))
        gf = 4.0 * B * A * (Np + 1) ** 2 * 64 / 1e9
        print(f"N={Np + 1:5d} B={B:2d} A={A}: small {ts:8.1f} us ({gf / ts * 1e-3:6.1f} TF/s)   large {tb:8.1f} us ({gf / tb * 1e-3:6.1f} TF/s)", flush=True)
