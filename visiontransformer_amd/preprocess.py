"""Pre-processing on the device.

Mirrors what the reference's scripts do on the host with torchvision/Pillow before the model sees anything:

* images : `transforms.Compose([transforms.Resize((S, S)), transforms.ToTensor()])` on a PIL RGB image
           (model/CE/trainCurrentViTmodel.py:48-51, model/CE/testViTModel.py:92-97) -> `Preprocessor.images()`
* masks  : `transforms.Resize((256, 256), NEAREST)` on the 'L' mask, `np.vectorize(value_to_class.get)`,
           `torch.tensor(..., dtype=torch.long)` (model/CE/classes.py:76-89) and, in training,
           `F.interpolate(y[:, None].float(), size, mode='nearest')` (classes.py:273-274) -> `Preprocessor.masks()`

Both run as HIP kernels of libvitseg and are bit-exact with Pillow / torch (tests/test_gpu_preproc.py); only the
decoded uint8 pixels cross PCIe (3 B/pixel instead of 12).  The per-axis coefficient / index tables are made by the
library's host functions once per source size and cached on the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib

NEAREST_PIL, NEAREST_TORCH = 0, 1


def resize_tables(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(taps, bounds int32 [out, 2], kk int32 [out, taps]) of one axis, from the library's host function."""
    taps = _lib.lib().vitseg_resize_taps(in_size, out_size)
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, taps), np.int32)
    _lib.check(_lib.lib().vitseg_resize_coeffs(in_size, out_size, bounds.ctypes.data, kk.ctypes.data))
    return taps, bounds, kk


def nearest_table(in_size: int, out_size: int, mode: int) -> np.ndarray:
    idx = np.zeros(out_size, np.int32)
    _lib.check(_lib.lib().vitseg_nearest_index(in_size, out_size, mode, idx.ctypes.data))
    return idx


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


class Preprocessor:
    """Device-side Resize + ToTensor for images and nearest-resize + remap for masks, for one model input size."""

    def __init__(self, image_size: int, device="cuda:0"):
        self.S = int(image_size)
        self.device = torch.device(device)
        self._axis: Dict[Tuple[int, int], tuple] = {}
        self._near: Dict[Tuple[int, int, int], torch.Tensor] = {}

    def _axis_tables(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        if key not in self._axis:
            taps, b, k = resize_tables(in_size, out_size)
            self._axis[key] = (taps, torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device),
                               int(b[0, 0]), int(b[-1, 0] + b[-1, 1]))
        return self._axis[key]

    def _nearest(self, in_size: int, out_size: int, mode: int) -> torch.Tensor:
        key = (in_size, out_size, mode)
        if key not in self._near:
            self._near[key] = torch.from_numpy(nearest_table(in_size, out_size, mode)).to(self.device)
        return self._near[key]

    def images(self, img: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """uint8 RGB [H, W, 3] or [n, H, W, 3] (any device; moved to the GPU as bytes) -> float32 [n, 3, S, S] in [0, 1]."""
        if img.dtype != torch.uint8 or img.shape[-1] != 3 or img.dim() not in (3, 4):
            raise ValueError(f"expected uint8 [H, W, 3] or [n, H, W, 3], got {img.dtype} {tuple(img.shape)}")
        if img.dim() == 3:
            img = img[None]
        img = img.to(self.device, non_blocking=True).contiguous()
        n, H, W, _ = img.shape
        S = self.S
        if out is None:
            out = torch.empty((n, 3, S, S), dtype=torch.float32, device=self.device)
        xt = self._axis_tables(W, S) if W != S else None
        yt = self._axis_tables(H, S) if H != S else None
        first, last = (yt[3], yt[4]) if yt else (0, H)
        tmp = torch.empty((n, last - first, S, 3), dtype=torch.uint8, device=self.device) if xt else None
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().vitseg_preprocess_u8(
                img.data_ptr(), n, H, W, S, _ptr(xt[1]) if xt else None, _ptr(xt[2]) if xt else None, xt[0] if xt else 0,
                _ptr(yt[1]) if yt else None, _ptr(yt[2]) if yt else None, yt[0] if yt else 0, first, last - first,
                _ptr(tmp), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return out

    def targets(self, y: torch.Tensor, size: Tuple[int, int], dtype=torch.long) -> torch.Tensor:
        """`F.interpolate(y[:, None].float(), size=size, mode='nearest').squeeze(1).long()` of class-index maps
        [n, H, W] (torch.long as the reference's datasets deliver them, or torch.uint8) -- LightningViTModel._resize_target,
        model/CE/classes.py:273-274 -- as ONE gather kernel on the device: no float round trip, no ATen kernels.
        `dtype` torch.long (the reference's result type) or torch.uint8 (what the fused CE loss reads)."""
        if y.dim() != 3 or y.dtype not in (torch.long, torch.uint8):
            raise ValueError(f"expected int64/uint8 [n, H, W] class indices, got {y.dtype} {tuple(y.shape)}")
        if dtype not in (torch.long, torch.uint8):
            raise ValueError("dtype must be torch.long or torch.uint8")
        y = y.to(self.device, non_blocking=True).contiguous()
        n, H, W = y.shape
        oh, ow = size
        out = torch.empty((n, oh, ow), dtype=dtype, device=self.device)
        yi, xi = self._nearest(H, oh, NEAREST_TORCH), self._nearest(W, ow, NEAREST_TORCH)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            if y.dtype == torch.long:
                _lib.check(_lib.lib().vitseg_resize_nearest_i64(y.data_ptr(), n, H, W, yi.data_ptr(), xi.data_ptr(), oh, ow,
                                                                int(dtype == torch.long), out.data_ptr(), st))
            else:
                _lib.check(_lib.lib().vitseg_resize_nearest_u8(y.data_ptr(), n, H, W, yi.data_ptr(), xi.data_ptr(), oh, ow,
                                                               None, int(dtype == torch.long), out.data_ptr(), st))
        return out

    def masks(self, mask: torch.Tensor, size: Tuple[int, int], mode: int = NEAREST_PIL,
              value_to_class: Optional[dict] = None, dtype=torch.long) -> torch.Tensor:
        """uint8 [H, W] or [n, H, W] label images -> [n, size] class indices (`dtype` torch.long as the reference's
        datasets produce, or torch.uint8).  mode: NEAREST_PIL for `transforms.Resize(..., NEAREST)`, NEAREST_TORCH for
        `F.interpolate(mode='nearest')`; `value_to_class` is the dataset's grey value -> class index mapping."""
        if mask.dtype != torch.uint8 or mask.dim() not in (2, 3):
            raise ValueError(f"expected uint8 [H, W] or [n, H, W], got {mask.dtype} {tuple(mask.shape)}")
        if dtype not in (torch.long, torch.uint8):
            raise ValueError("dtype must be torch.long or torch.uint8")
        if mask.dim() == 2:
            mask = mask[None]
        mask = mask.to(self.device, non_blocking=True).contiguous()
        n, H, W = mask.shape
        oh, ow = size
        lut = None
        if value_to_class is not None:
            table = np.zeros(256, np.uint8)
            for v, c in value_to_class.items():
                table[int(v)] = int(c)
            lut = torch.from_numpy(table).to(self.device)
        out = torch.empty((n, oh, ow), dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().vitseg_resize_nearest_u8(
                mask.data_ptr(), n, H, W, self._nearest(H, oh, mode).data_ptr(), self._nearest(W, ow, mode).data_ptr(),
                oh, ow, _ptr(lut), int(dtype == torch.long), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return out
