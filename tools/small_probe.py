#!/usr/bin/env python3
"""Times the small-batch fp32 GEMM (csrc/gemm_f32s.hip) per layer shape, tile variant and row count:
us per launch (200 back-to-back launches on one stream) beside the fp32-MFMA-peak time of the same FLOPs.
    python tools/small_probe.py [rows ...]        (default rows: 197 788 1576)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
rows = [int(a) for a in sys.argv[1:]] or [197, 788, 1576]
D, I = 768, 3072
shapes = [("qkv", 3 * D, D, 0), ("fc1", I, D, 1), ("o_proj", D, D, 2), ("fc2", D, I, 2)]


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for M in rows:
    for name, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        C = torch.empty(M, N, device=dev)
        S = L.vitseg_small_splits(N, K)
        scratch = torch.empty(S * M * N, device=dev)
        X, H = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
        lnw, lnb = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        if epi == 2:
            fn = lambda: _lib.check(L.vitseg_op_linear_resln_f32_small(A.data_ptr(), W.data_ptr(), bias.data_ptr(), X.data_ptr(), lnw.data_ptr(),
                                                                      lnb.data_ptr(), H.data_ptr(), scratch.data_ptr(), scratch.numel(), M, N, K, 1e-12, st()))
        else:
            fn = lambda: _lib.check(L.vitseg_op_linear_f32_small(A.data_ptr(), W.data_ptr(), bias.data_ptr(), C.data_ptr(), M, N, K, epi, st()))
        ideal = 2.0 * M * N * K / 157.3e12 * 1e6
        res = []
        for v in range(0, 6):
            with _lib.option("small_variant", v):
                res.append(timed(fn))
        print(f"M={M:5d} {name:7s} N={N:5d} K={K:5d} S={S}  peak-time {ideal:6.1f} us | plan {res[0]:6.1f} | " +
              " ".join(f"v{v}:{t:6.1f}" for v, t in enumerate(res[1:], 1)) + ("   (+ the row kernel)" if epi == 2 else ""), flush=True)
