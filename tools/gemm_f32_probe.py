#!/usr/bin/env python3
"""A/B of the fp32 linear kernels on the model's shapes, interleaved rounds in ONE process (random operands): the
persistent 256x128 kernel (csrc/gemm_f32p.hip) against gemm.hip's 128x128 tile kernel (VITSEG_NO_F32P=1); outputs must
be bit-identical.    python3 tools/gemm_f32_probe.py [--batch 32] [--rounds 3] [--gn N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
L = _lib.lib()
M = a.batch * 1024
SHAPES = [("qkv", 2304, 768, 0), ("fc1+gelu", 3072, 768, 1), ("fc2+res", 768, 3072, 2), ("o_proj+res", 768, 768, 2)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e-3


for name, N, K, epi in SHAPES:
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) * 0.1
    R0 = torch.randn(M, N, device=dev)
    outs, best = {}, {}
    for rnd in range(a.rounds + 1):
        for label, env in (("f32p", None), ("tile", "1")):
            _lib.set_option("no_f32p", 1 if env else 0)
            C = R0.clone() if epi == 2 else torch.zeros(M, N, device=dev)

            def fn():
                _lib.check(L.vitseg_op_linear_f32(A.data_ptr(), W.data_ptr(), b.data_ptr(), C.data_ptr() if epi == 2 else None,
                                                  C.data_ptr(), M, N, K, epi, st))
            if rnd == 0:
                fn()
                torch.cuda.synchronize()
                outs[label] = C.clone()
                continue
            t = timed(fn)
            best[label] = min(best.get(label, 1e9), t)
    same = torch.equal(outs["f32p"], outs["tile"])
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M} N={N} K={K}: f32p {best['f32p']*1e3:7.3f} ms = {fl/best['f32p']/1e12:6.1f} TF/s | "
          f"tile {best['tile']*1e3:7.3f} ms = {fl/best['tile']/1e12:6.1f} TF/s | bitwise equal: {same}", flush=True)
