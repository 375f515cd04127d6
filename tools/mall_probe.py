#!/usr/bin/env python3
"""Does a working set that fits the 256 MB Infinity Cache re-read faster than HBM?  Read-only passes (torch's sum
reduction as the reader) over buffers of growing size; the rate drops from ~5 TB/s to the HBM rate once the buffer no
longer fits (profiles/r02_mall_probe.txt)."""
import time
import torch
dev = "cuda:0"
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    x = torch.ones(mb * 1024 * 1024 // 4, device=dev)
    for _ in range(3):
        x.sum()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(4, 4096 // mb)
    e0.record()
    for _ in range(n):
        x.sum()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"re-read {mb:5d} MB: {ms * 1e3:8.1f} us  {mb * 1.048576 / ms:8.0f} GB/s")
