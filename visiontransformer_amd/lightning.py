"""`LightningViTModel` -- the CE training/eval module of the reference without the `lightning`
dependency (/root/reference/model/CE/classes.py:264-297).

Same constructor, `forward`, `_resize_target`, `training_step`, `validation_step`,
`configure_optimizers` (Adam lr=1e-5), and the same checkpoint layout: `state_dict()` keys carry the
`model.` prefix of the reference's attribute name (classes.py:267) and `load_state_dict` accepts
`torch.load(ckpt)['state_dict']` as written by Lightning (model/CE/testViTModel.py:117-118).
Logging (`self.log`) is replaced by a plain `logged` dict; a minimal trainer loop lives in
`visiontransformer_amd.trainer`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import ViTSegmentationModel


class LightningViTModel(nn.Module):
    def __init__(self, num_classes, patch_size, hidden_size, num_hidden_layers, num_attention_heads, **kw):
        super().__init__()
        self.model = ViTSegmentationModel(num_classes, patch_size, hidden_size, num_hidden_layers,
                                          num_attention_heads, **kw)
        self.logged = {}

    def forward(self, x):
        return self.model(x)

    def _resize_target(self, y, size, dtype=torch.long):
        """classes.py:273-274: F.interpolate(y[:, None].float(), size, mode='nearest') -> long (idx = min(floor(dst*in/out),
        in-1)).  Targets on the GPU go through one gather kernel (preprocess.Preprocessor.targets: no float round trip, no
        ATen kernels in the training step); a CPU tensor is host-side label preprocessing as in the reference."""
        if y.is_cuda and y.dim() == 3 and y.dtype in (torch.long, torch.uint8):
            from .preprocess import Preprocessor
            if getattr(self, "_prep", None) is None or self._prep.device != y.device:
                self._prep = Preprocessor(self.model.cfg.image_size, device=y.device)
            return self._prep.targets(y, tuple(size), dtype=dtype)
        return F.interpolate(y.unsqueeze(1).float(), size=size, mode="nearest").squeeze(1).to(dtype)

    def _loss(self, batch, grad_scale=None):
        x, y = batch
        S = self.model.cfg.image_size  # the reference hard-codes (224, 224) = its image_size (classes.py:278)
        # uint8 class indices: what the fused CE kernels read (a quarter of the int64 bytes); C <= 32 in training
        y = self._resize_target(y.to(x.device, non_blocking=True), size=(S, S), dtype=torch.uint8)
        return self.model.ce_loss(x, y, grad_scale=grad_scale)

    # `logged` holds DEVICE scalars: reading one (float(...)) is the only host sync, and only the caller decides when
    def training_step(self, batch, batch_idx, grad_scale=None):
        loss = self._loss(batch, grad_scale)
        self.logged["train_loss"] = loss.detach()
        return loss

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            loss = self._loss(batch)
        self.logged["valid_loss"] = loss
        return loss

    def configure_optimizers(self):
        from .optim import FusedAdam
        return FusedAdam(self.parameters(), lr=1e-5)  # Adam(lr=1e-5), classes.py:296-297

    # checkpoint schema: every key prefixed with "model." like the reference module tree
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        return self.model.state_dict(destination=destination, prefix=prefix + "model.", keep_vars=keep_vars)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        return self.model.load_state_dict(state_dict, strict=strict)
