#!/usr/bin/env python3
"""Times the training forms of the small-batch fp32 GEMM (csrc/gemm_f32s.hip: activation gradients with the weight in T-form,
weight gradients with both operands token-major) per layer shape and tile variant: us per launch beside the planner's choice.
    python tools/train_gemm_probe.py [rows]      (default 788: the reference's batch 4 x 224x224)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 788
D, I = 768, 3072


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def row(name, fn, flops):
    res = []
    for v in range(0, 6):
        with _lib.option("small_variant", v):
            res.append(timed(fn))
    print(f"M={M:5d} {name:28s} peak-time {flops / 157.3e12 * 1e6:6.1f} us | plan {res[0]:6.1f} | " +
          " ".join(f"v{v}:{t:6.1f}" for v, t in enumerate(res[1:], 1)), flush=True)


# activation gradients dX[M, Kd] = dY[M, Nd] . W[Nd, Kd]
for name, Nd, Kd, epi in (("dgrad fc2 (dGELU)", D, I, 5), ("dgrad fc1", I, D, 0), ("dgrad o_proj", D, D, 0), ("dgrad qkv", 3 * D, D, 0)):
    dY = torch.randn(M, Nd, device=dev)
    W = torch.randn(Nd, Kd, device=dev) * 0.05
    R = torch.randn(M, Kd, device=dev)
    dX = torch.empty(M, Kd, device=dev)
    S = L.vitseg_small_splits(Kd, Nd)
    scr = torch.empty(max(S, 1) * M * Kd, device=dev)
    fn = lambda: _lib.check(L.vitseg_op_dgrad_f32_small(dY.data_ptr(), W.data_ptr(), R.data_ptr(), dX.data_ptr(), scr.data_ptr(), scr.numel(),
                                                        M, Nd, Kd, epi, st()))
    row(name + (f" [{S} slabs + sum]" if S > 1 else ""), fn, 2.0 * M * Nd * Kd)
# weight gradients dW[Nd, Kd] = dY[M, Nd]^T . X[M, Kd]
for name, Nd, Kd in (("wgrad fc2", D, I), ("wgrad fc1", I, D), ("wgrad o_proj", D, D), ("wgrad qkv", 3 * D, D)):
    dY = torch.randn(M, Nd, device=dev)
    X = torch.randn(M, Kd, device=dev)
    dW = torch.empty(Nd, Kd, device=dev)
    fn = lambda: _lib.check(L.vitseg_op_wgrad_f32_small(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), M, Nd, Kd, st()))
    row(name, fn, 2.0 * M * Nd * Kd)
