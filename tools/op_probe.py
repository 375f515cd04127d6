#!/usr/bin/env python3
"""Launches one operator of libvitseg.so repeatedly (for rocprofv3 --pmc / --kernel-trace runs).

    python3 tools/op_probe.py linear --M 32800 --N 3072 --K 768 --epi 1 --iters 10
    python3 tools/op_probe.py attention --B 32 --Np 1024 --A 12
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("op", choices=["linear", "attention", "layernorm", "linear_ex", "wgrad", "attn_bwd"])
ap.add_argument("--M", type=int, default=32800)
ap.add_argument("--N", type=int, default=3072)
ap.add_argument("--K", type=int, default=768)
ap.add_argument("--epi", type=int, default=0)
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--Np", type=int, default=1024)
ap.add_argument("--A", type=int, default=12)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--drop", type=float, default=0.0)
ap.add_argument("--scale", type=float, default=1.0, help="attention: standard deviation of the q | k | v entries")
ap.add_argument("--aux", action="store_true", help="linear_ex, epi 1: also write the saved GELU derivative")
ap.add_argument("--words", action="store_true", help="attn_bwd: dropout keep bits as precomputed mask words")
a = ap.parse_args()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
if a.op == "attn_bwd":   # bf16 attention forward (with lse) + backward, optional attention dropout
    D = 64 * a.A
    Mt = a.B * a.Np + a.B
    qkv = torch.randn(Mt, 3 * D, device=dev).to(torch.bfloat16)
    dctx = torch.randn(Mt, D, device=dev).to(torch.bfloat16)
    ctx = torch.empty(Mt, D, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(a.B * a.A * (a.Np + 1), device=dev)
    scr = torch.empty(L.vitseg_attention_bwd_scratch_floats(a.B, a.Np, a.A), device=dev)
    dqkv = torch.empty(Mt, 3 * D, device=dev, dtype=torch.bfloat16)
    mw = torch.empty(L.vitseg_attention_dropmask_bytes(a.B, a.Np, a.A), dtype=torch.uint8, device=dev) if a.words else None
    run = lambda: _lib.check(L.vitseg_op_attention_bwd_bf16(qkv.data_ptr(), dctx.data_ptr(), ctx.data_ptr(), lse.data_ptr(),
                                                            scr.data_ptr(), dqkv.data_ptr(), a.B, a.Np, a.A, a.drop, 1234, 9,
                                                            mw.data_ptr() if mw is not None else None, None, st))
    work = 14.0 * a.B * a.A * (a.Np + 1) ** 2 * 64   # 2 + 5 products of 2 N^2 hd
elif a.op == "wgrad":   # dW[M,N] = dY^T X over K token rows, bf16 operands token-major
    dY = torch.randn(a.K, a.M, device=dev).to(torch.bfloat16)
    X = torch.randn(a.K, a.N, device=dev).to(torch.bfloat16)
    dW = torch.empty(a.M, a.N, device=dev)
    zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
    scratch = torch.empty(L.vitseg_op_wgrad_bf16_scratch_floats(a.M, a.N, a.K), device=dev)
    run = lambda: _lib.check(L.vitseg_op_wgrad_bf16(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), scratch.data_ptr(),
                                                    zeros.data_ptr(), a.M, a.N, a.K, st))
    work = 2.0 * a.M * a.N * a.K
elif a.op == "linear_ex":   # 16-bit linear with every epilogue (0 bias, 1 GELU, 2 residual fp32, 5 dGELU), bf16
    A = torch.randn(a.M, a.K, device=dev).to(torch.bfloat16)
    W = (torch.randn(a.N, a.K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(a.N, device=dev)
    R = torch.randn(a.M, a.N, device=dev).to(torch.float32 if a.epi == 2 else torch.bfloat16) if a.epi in (2, 5) else None
    C = R if a.epi == 2 else torch.zeros(a.M, a.N, device=dev, dtype=torch.bfloat16)
    aux = torch.zeros(a.M, a.N, device=dev, dtype=torch.bfloat16) if a.aux else None
    run = lambda: _lib.check(L.vitseg_op_linear_h16_ex(A.data_ptr(), W.data_ptr(), b.data_ptr() if a.epi != 5 else None,
                                                       R.data_ptr() if R is not None else None, C.data_ptr(),
                                                       aux.data_ptr() if a.aux else None, a.M, a.N,
                                                       a.K, a.epi, 0, 0, None, 0, 0.0, 0, 0, None, None, st))
    work = 2.0 * a.M * a.N * a.K
elif a.op == "linear" and a.bf16:
    A = torch.randn(a.M, a.K, device=dev).to(torch.bfloat16)
    W = (torch.randn(a.N, a.K, device=dev) * 0.05).to(torch.bfloat16)
    b = torch.randn(a.N, device=dev)
    C = torch.zeros(a.M, a.N, device=dev, dtype=torch.float32 if a.epi == 2 else torch.bfloat16)
    run = lambda: _lib.check(L.vitseg_op_linear_bf16(A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr(), C.data_ptr(),
                                                     a.M, a.N, a.K, a.epi, st))
    work = 2.0 * a.M * a.N * a.K
elif a.op == "attention" and a.bf16:
    D = 64 * a.A
    qkv = (torch.randn(a.B * a.Np + a.B, 3 * D, device=dev) * a.scale).to(torch.bfloat16)
    ctx = torch.empty(a.B * a.Np + a.B, D, device=dev, dtype=torch.bfloat16)
    run = lambda: _lib.check(L.vitseg_op_attention_bf16(qkv.data_ptr(), ctx.data_ptr(), a.B, a.Np, a.A, st))
    work = 4.0 * a.B * a.A * (a.Np + 1) ** 2 * 64
elif a.op == "linear":
    A = torch.randn(a.M, a.K, device=dev)
    W = torch.randn(a.N, a.K, device=dev) * 0.05
    b = torch.randn(a.N, device=dev)
    C = torch.zeros(a.M, a.N, device=dev)
    run = lambda: _lib.check(L.vitseg_op_linear_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr(), C.data_ptr(),
                                                    a.M, a.N, a.K, a.epi, st))
    work = 2.0 * a.M * a.N * a.K
elif a.op == "attention":
    D = 64 * a.A
    qkv = torch.randn(a.B * a.Np + a.B, 3 * D, device=dev)
    ctx = torch.empty(a.B * a.Np + a.B, D, device=dev)
    run = lambda: _lib.check(L.vitseg_op_attention_f32(qkv.data_ptr(), ctx.data_ptr(), a.B, a.Np, a.A, st))
    work = 4.0 * a.B * a.A * (a.Np + 1) ** 2 * 64
else:
    x = torch.randn(a.M, a.K, device=dev)
    w = torch.randn(a.K, device=dev)
    y = torch.empty_like(x)
    run = lambda: _lib.check(L.vitseg_op_layernorm_f32(x.data_ptr(), w.data_ptr(), w.data_ptr(), y.data_ptr(), a.M, a.K,
                                                       1e-12, st))
    work = 2.0 * a.M * a.K * 4
run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print(f"{a.op}: {dt * 1e3:.3f} ms/launch, {work / dt / 1e12:.2f} T(FLOP|B)/s")
