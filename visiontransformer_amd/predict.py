"""`predict()` -- the worker-side call the reference's job dispatcher expects but does not contain.

The reference only ships the single-image script model/CE/testViTModel.py:92-126 (PIL -> Resize ->
ToTensor -> model.eval() -> logits.sigmoid() -> argmax) and a Django view that POSTs the image to an
external orchestrator (backend/core/views.py:97-114).  `predict()` is that script's inference body
as a function: image in, uint8 class-index mask out, computed by libvitseg on the MI355X (the
sigmoid->argmax is fused into the decoder-tail kernel).
"""
from __future__ import annotations

import io
import os
from typing import Optional, Union

import numpy as np
import torch

from .lightning import LightningViTModel
from .model import ViTSegmentationModel

# the nine (patch, hidden, layers, heads) configurations of the reference (testViTModel.py:73-83)
CONFIGURATIONS = {
    0: (16, 768, 12, 12), 1: (16, 512, 8, 8), 2: (16, 1024, 16, 16),
    3: (8, 512, 8, 8), 4: (8, 768, 12, 12), 5: (8, 1024, 16, 16),
    6: (4, 512, 8, 8), 7: (4, 768, 12, 12), 8: (4, 1024, 16, 16),
}


def load_model(model_id_or_config, num_classes: int, checkpoint: Optional[str] = None, *, image_size: int = 224,
               precision: str = "fp32", device="cuda:0") -> LightningViTModel:
    """Builds `LightningViTModel` for a reference configuration ID (or a (P, D, L, A) tuple) and loads a
    Lightning checkpoint `{'state_dict': ...}` if given (testViTModel.py:109-119)."""
    P, D, L, A = CONFIGURATIONS[model_id_or_config] if isinstance(model_id_or_config, int) else model_id_or_config
    model = LightningViTModel(num_classes, P, D, L, A, image_size=image_size, precision=precision, device=device)
    if checkpoint is not None:
        ck = torch.load(checkpoint, map_location="cpu")
        model.load_state_dict(ck["state_dict"] if "state_dict" in ck else ck)
    return model.eval()


def decode(image) -> np.ndarray:
    """PIL image / path / encoded bytes -> uint8 RGB [H, W, 3] (`Image.open(...).convert('RGB')`, classes.py:72)."""
    from PIL import Image
    if isinstance(image, (bytes, bytearray)):
        image = Image.open(io.BytesIO(image))
    elif isinstance(image, (str, os.PathLike)):
        image = Image.open(image)
    if isinstance(image, np.ndarray):
        return np.ascontiguousarray(image, dtype=np.uint8)
    return np.array(image.convert("RGB"), dtype=np.uint8)   # a writable copy (torch.from_numpy wants one)


_PRE = {}


def preprocess(image, size: int, device="cuda:0") -> torch.Tensor:
    """image -> float32 [1, 3, size, size] in [0, 1] on `device`: transforms.Resize((S, S)) [Pillow's antialiased
    bilinear] + ToTensor (testViTModel.py:92-97), computed by libvitseg bit-exactly (preprocess.Preprocessor); only the
    decoded bytes are uploaded."""
    from .preprocess import Preprocessor
    key = (int(size), str(device))
    if key not in _PRE:
        _PRE[key] = Preprocessor(size, device)
    return _PRE[key].images(torch.from_numpy(decode(image)))


def predict(image, model: Union[LightningViTModel, ViTSegmentationModel], *, index_to_color=None,
            return_logits: bool = False):
    """uint8 class-index mask [S, S] (numpy) for one image; optionally also an RGB rendering
    `index_to_color[mask]` (testViTModel.py:139-143) and/or the fp32 logits [C, S, S]."""
    seg = model.model if isinstance(model, LightningViTModel) else model
    x = preprocess(image, seg.cfg.image_size, seg.arena.device)
    out = seg.predict_mask(x, return_logits=return_logits)
    mask = (out[0] if return_logits else out)[0].cpu().numpy()
    res = [mask]
    if index_to_color is not None:
        res.append(np.asarray(index_to_color, dtype=np.uint8)[mask])
    if return_logits:
        res.append(out[1][0].cpu().numpy())
    return res[0] if len(res) == 1 else tuple(res)
