#!/usr/bin/env python3
"""A/B of the 16-bit GEMM kernels on the model's shapes, interleaved rounds in ONE process (random operands):
the persistent 8-phase kernel (gemm_p8.hip) against the previous tile kernels (VITSEG_NO_P8=1) and, for calibration
only, the vendor GEMM behind torch.matmul.  Every variant is checked against an fp32 product of the same operands on a
sample of rows first.    python3 tools/gemm_probe.py [--batch 64] [--rounds 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--fmt", default="bf16")
a = ap.parse_args()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
dt = torch.bfloat16 if a.fmt == "bf16" else torch.float16
M = a.batch * 1024           # the whole-tile body (the CLS rows go through the split-K side launch)
# (name, N, K, epilogue): forward and dgrad GEMMs of one ViT-B/16 layer
SHAPES = [("qkv", 2304, 768, 0), ("fc1+gelu", 3072, 768, 1), ("fc2+res", 768, 3072, 2), ("o_proj+res", 768, 768, 2),
          ("dgrad fc2 (dgelu)", 3072, 768, 5), ("dgrad fc1", 768, 3072, 0), ("dgrad qkv", 768, 2304, 0)]


def call(A, W, b, R, C, N, K, epi):
    _lib.check(L.vitseg_op_linear_h16_ex(A.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None,
                                         R.data_ptr() if R is not None else None, C.data_ptr(), None, M, N, K, epi,
                                         int(a.fmt != "bf16"), 0, None, 0, 0.0, 0, 0, None, None, st))


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e-3


for name, N, K, epi in SHAPES:
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    b = torch.randn(N, device=dev) * 0.1 if epi != 5 else None
    out_dt = torch.float32 if epi == 2 else dt
    R = torch.randn(M, N, device=dev).to(torch.float32 if epi == 2 else dt) if epi in (2, 5) else None
    outs = {}
    for label, env in (("p8", None), ("old", "1")):
        _lib.set_option("no_p8", 1 if env else 0)
        C = R.clone() if epi == 2 else torch.zeros(M, N, device=dev, dtype=out_dt)
        call(A, W, b, C if epi == 2 else R, C, N, K, epi)
        outs[label] = C.float()
    torch.cuda.synchronize()
    rows = torch.randint(0, M, (512,), device=dev)
    acc = A[rows].float() @ W.float().T
    if epi == 0:
        ref = acc + b
    elif epi == 1:
        ref = torch.nn.functional.gelu(acc + b)
    elif epi == 2:
        ref = R[rows] + acc + b
    else:
        u = R[rows].float()
        ref = acc * (0.5 * (1 + torch.erf(u / 2 ** 0.5)) + u * torch.exp(-0.5 * u * u) / (2 * 3.141592653589793) ** 0.5)
    err = {k: float((v[rows] - ref).abs().max()) for k, v in outs.items()}
    same = float((outs["p8"] - outs["old"]).abs().max())
    fl = 2.0 * M * N * K
    best = {"p8": 1e9, "old": 1e9, "vendor": 1e9}
    C = R.clone() if epi == 2 else torch.zeros(M, N, device=dev, dtype=out_dt)
    for _ in range(a.rounds):
        _lib.set_option("no_p8", 0)
        best["p8"] = min(best["p8"], timed(lambda: call(A, W, b, C if epi == 2 else R, C, N, K, epi)))
        _lib.set_option("no_p8", 1)
        best["old"] = min(best["old"], timed(lambda: call(A, W, b, C if epi == 2 else R, C, N, K, epi)))
        best["vendor"] = min(best["vendor"], timed(lambda: torch.matmul(A, W.t())))
    _lib.set_option("no_p8", 0)
    print(f"{name:18s} M={M} N={N} K={K}: p8 {best['p8'] * 1e6:7.1f} us {fl / best['p8'] / 1e12:7.1f} TF/s | old "
          f"{best['old'] * 1e6:7.1f} us {fl / best['old'] / 1e12:7.1f} | vendor plain {fl / best['vendor'] / 1e12:7.1f} | "
          f"max err p8 {err['p8']:.3e} old {err['old']:.3e} p8-old {same:.3e}", flush=True)


# ---- weight gradients: dW[M_out, N_out] = dY^T X over the token rows (both operands token-major) ----
print("weight gradients (T-form x T-form, split over the token rows):")
Kt = M + a.batch
zeros = torch.zeros(256, dtype=torch.uint8, device=dev)
for name, Mo, No in [("dW fc2", 768, 3072), ("dW fc1", 3072, 768), ("dW o_proj", 768, 768), ("dW qkv", 2304, 768)]:
    dY = torch.randn(Kt, Mo, device=dev).to(torch.bfloat16)
    X = torch.randn(Kt, No, device=dev).to(torch.bfloat16)
    dW = torch.empty(Mo, No, device=dev)
    scratch = torch.empty(L.vitseg_op_wgrad_bf16_scratch_floats(Mo, No, Kt), device=dev)
    fn = lambda: _lib.check(L.vitseg_op_wgrad_bf16(dY.data_ptr(), X.data_ptr(), dW.data_ptr(), scratch.data_ptr(),
                                                   zeros.data_ptr(), Mo, No, Kt, st))
    outs = {}
    for label, env in (("p8", None), ("old", "1")):
        _lib.set_option("no_p8", 1 if env else 0)
        fn()
        outs[label] = dW.clone()
    ref = dY[:, :64].float().T @ X.float()
    err = {k: float((v[:64] - ref).abs().max() / ref.abs().max()) for k, v in outs.items()}
    fl = 2.0 * Mo * No * Kt
    best = {"p8": 1e9, "old": 1e9, "vendor": 1e9}
    for _ in range(a.rounds):
        _lib.set_option("no_p8", 0)
        best["p8"] = min(best["p8"], timed(fn))
        _lib.set_option("no_p8", 1)
        best["old"] = min(best["old"], timed(fn))
        best["vendor"] = min(best["vendor"], timed(lambda: torch.matmul(dY.t(), X)))
    _lib.set_option("no_p8", 0)
    print(f"{name:18s} M={Mo} N={No} K={Kt}: p8 {best['p8'] * 1e6:7.1f} us {fl / best['p8'] / 1e12:7.1f} TF/s | old "
          f"{best['old'] * 1e6:7.1f} us {fl / best['old'] / 1e12:7.1f} | vendor plain {fl / best['vendor'] / 1e12:7.1f} | "
          f"rel err p8 {err['p8']:.2e} old {err['old']:.2e}", flush=True)
