"""numpy restatement of libvitseg's counter-based dropout (csrc/common.hpp: fmix32 / drop_key / drop_keep) that
produces the masks in the REFERENCE layout ([B, N, D] / [B, A, N, N] with the CLS token first), so the oracle can
be run with exactly the masks the HIP kernels use."""
import numpy as np
import torch


def fmix32(h):
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * np.uint32(0xC2B2AE35)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def mad24(a, k, c):
    """low 24 bits of a times the 24-bit constant k, plus c, mod 2^32 (v_mad_u32_u24)"""
    a, c = np.broadcast_arrays(np.asarray(a, dtype=np.uint32), np.asarray(c, dtype=np.uint32))
    return ((a.astype(np.uint64) & np.uint64(0xFFFFFF)) * np.uint64(k & 0xFFFFFF) + c.astype(np.uint64)).astype(np.uint32)


class Masks:
    def __init__(self, p, seed64, B, Np, A):
        self.p, self.B, self.Np, self.A = p, B, Np, A
        self.seed = np.uint32((seed64 ^ (seed64 >> 32)) & 0xFFFFFFFF)
        self.thresh = np.uint32(max(1, int(p * 65536.0 + 0.5)))
        self.scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))

    def _keep(self, stream, major, minor):
        with np.errstate(over="ignore"):
            key = fmix32(self.seed ^ (np.uint32(stream) * np.uint32(0x9E3779B1)) ^ (major.astype(np.uint32) * np.uint32(0x85EBCA77)))
            m = minor.astype(np.uint32)
            h = mad24(m >> np.uint32(1), 0x9E3779, key)
            h ^= h >> np.uint32(15)
            h = (h * np.uint32(0x85EBCA6B)).astype(np.uint32)
            return np.where(m & np.uint32(1), h >> np.uint32(16), h & np.uint32(0xFFFF)) >= self.thresh

    def _gen(self, n_ref):  # reference token index (CLS first) -> generic index (CLS last)
        return np.where(n_ref == 0, self.Np, n_ref - 1)

    def rows(self, layer, site, shape):
        B, N, D = shape
        b = np.arange(B)[:, None]
        n = self._gen(np.arange(N))[None, :]
        row = np.where(n == self.Np, B * self.Np + b, b * self.Np + n)          # patches-first row index
        keep = self._keep(layer * 8 + site, row[:, :, None], np.arange(D)[None, None, :])
        return torch.from_numpy(np.where(keep, self.scale, np.float32(0)).astype(np.float32))

    def attn(self, layer, shape):
        B, A, N, _ = shape
        bh = (np.arange(B)[:, None] * A + np.arange(A)[None, :])[:, :, None, None]
        q = self._gen(np.arange(N))[None, None, :, None]
        k = self._gen(np.arange(N))[None, None, None, :]
        keep = self._keep(layer * 8 + 1, bh * N + q, np.broadcast_to(k, (B, A, N, N)))
        return torch.from_numpy(np.where(keep, self.scale, np.float32(0)).astype(np.float32))
