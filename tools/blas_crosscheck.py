#!/usr/bin/env python3
"""Calibration only (never on the product path): what the vendor GEMM (hipBLASLt behind torch.matmul) reaches on the
model's shapes, next to libvitseg's kernels.  python3 tools/blas_crosscheck.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
M = 32800


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for N, K, name in [(3072, 768, "fc1"), (2304, 768, "qkv"), (768, 3072, "fc2"), (768, 768, "o_proj")]:
    fl = 2.0 * M * N * K
    for dt, label in [(torch.bfloat16, "bf16"), (torch.float16, "fp16"), (torch.float32, "fp32")]:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        t_blas = timeit(lambda: torch.matmul(A, W.t()))
        b = torch.zeros(N, device=dev)
        C = torch.empty(M, N, device=dev, dtype=dt)
        if dt == torch.float32:
            t_own = timeit(lambda: _lib.check(L.vitseg_op_linear_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), None, C.data_ptr(), M, N, K, 0, st)))
            t_x3 = timeit(lambda: _lib.check(L.vitseg_op_linear_f32x3(A.data_ptr(), W.data_ptr(), b.data_ptr(), None, C.data_ptr(), M, N, K, 0, st)))
            print(f"{name:7s} {label}: vendor {fl / t_blas / 1e12:7.1f} TF/s | libvitseg fp32 MFMA {fl / t_own / 1e12:7.1f} | fp32x3 {fl / t_x3 / 1e12:7.1f}")
        else:
            fn = L.vitseg_op_linear_bf16 if dt == torch.bfloat16 else L.vitseg_op_linear_f16
            t_own = timeit(lambda: _lib.check(fn(A.data_ptr(), W.data_ptr(), b.data_ptr(), None, C.data_ptr(), M, N, K, 0, st)))
            print(f"{name:7s} {label}: vendor {fl / t_blas / 1e12:7.1f} TF/s | libvitseg {fl / t_own / 1e12:7.1f}")
