#!/usr/bin/env python3
"""Where a small-batch GEMM launch spends its time: per-block stamps written by the kernel itself (csrc/small.hpp
SGemm::stamps) -- block start skew, prologue (first operands landed), K loop, epilogue, and how the dispatcher placed the
blocks on CUs.  python tools/small_stamps.py [M N K [lds_pad_bytes]]"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visiontransformer_amd import _lib  # noqa: E402

dev = "cuda:0"
L = _lib.lib()
L.vitseg_dbg_linear_f32_small.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p]
args = [int(a) for a in sys.argv[1:]]
M, N, K = (args + [197, 2304, 768])[:3] if len(args) >= 3 else (197, 2304, 768)
pad = args[3] if len(args) > 3 else 0
A, W, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
Cc = torch.empty(M, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
for v in range(1, 6):
    stamps = torch.zeros(8 * 4096, dtype=torch.int64, device=dev)
    with _lib.option("small_variant", v):
        for _ in range(3):   # the last launch's stamps are read (warm code, cold-ish operands do not matter here)
            stamps.zero_()
            _lib.check(L.vitseg_dbg_linear_f32_small(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cc.data_ptr(), M, N, K, 0, stamps.data_ptr(), pad, st))
        torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    nb = len(s)
    t0 = s[:, 0].min()
    start_us = (s[:, 0] - t0) / 100.0
    end_us = (s[:, 1] - t0) / 100.0
    pro, loop, epi = s[:, 3] - s[:, 2], s[:, 4] - s[:, 3], s[:, 5] - s[:, 4]
    place = collections.Counter((int(r[7]) & 7, (int(r[6]) >> 13) & 7, (int(r[6]) >> 12) & 1, (int(r[6]) >> 8) & 15) for r in s)
    per_cu = collections.Counter(place.values())
    clk = np.median((s[:, 5] - s[:, 2]) / np.maximum((s[:, 1] - s[:, 0]), 1)) * 100e6 / 1e9
    print(f"M={M} N={N} K={K} variant {v} pad {pad}: {nb} blocks on {len(place)} CUs (blocks per CU: {dict(per_cu)}), span {end_us.max():.1f} us, "
          f"starts {np.median(start_us):.1f}/{start_us.max():.1f} us (median/max), clock {clk:.2f} GHz\n"
          f"    cycles median (max): prologue {int(np.median(pro))} ({pro.max()}), K loop {int(np.median(loop))} ({loop.max()}), epilogue {int(np.median(epi))} ({epi.max()})", flush=True)
