"""PAED entry points of the reference (model/PAED/classes.py) on the MI355X model.

`ViTSegmentationModel` is the same class as in model/CE (the reference keeps a byte-identical copy at
model/PAED/classes.py:372-413); what differs is the loss tail.  Both tails and their
gradients are libvitseg kernels: the 17-class soft PAED loss (`paed_multiclass_loss_fused`, csrc/paed_loss.hip) and the
binary trainer's sigmoid + BCE + Dice + SDF-resize + Sobel + per-image max-normalised edge term
(`paed_binary_loss_fused`, csrc/paed_binary.hip).  Either way the gradient reaches the parameters through libvitseg's
backward (`vitseg_backward` with `grad_logits`).  Lightning's logging is replaced by a `logged` dict.  The plain-torch functions below
(`paed_loss_multiclass_soft`, ...) mirror the reference's free functions of the same names.

  * `LightningViTModel` -- 17-class soft-PAED loss, Adam(lr=1e-4)          (model/PAED/classes.py:415-487)
  * `PAEDTrainer`       -- binary BCE + 0.1 Dice + 5 |soft-PAED|, AdamW(1e-4) + ReduceLROnPlateau (:490-701)
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .model import ViTSegmentationModel


def paed_loss_multiclass_soft(msk, pred_mask, num_classes=17, sigma=3, class_penalty=True):
    """Soft PAED for C classes (model/PAED/classes.py:336-369): both the one-hot target and the predicted
    probabilities are blurred per class with a normalised (6 sigma + 1)^2 Gaussian; the L1 difference of the
    blurred maps, weighted by 2 * target * (1 - prob) when `class_penalty`, is averaged over space, then
    classes, then the batch."""
    C = msk.shape[1]
    k = int(6 * sigma + 1)
    ax = torch.arange(k, dtype=torch.float32) - k // 2
    g1 = torch.exp(-(ax ** 2) / (2 * sigma ** 2))
    g2 = g1[:, None] * g1[None, :]
    g2 = (g2 / g2.sum()).to(msk.device)[None, None].repeat(C, 1, 1, 1)
    blur_t = F.conv2d(msk, g2, padding=k // 2, groups=C)
    blur_p = F.conv2d(pred_mask, g2, padding=k // 2, groups=C)
    diff = (blur_t - blur_p).abs()
    if class_penalty:
        diff = msk * (1 - pred_mask) * diff * 2
    return diff.mean(dim=[2, 3]).mean(dim=1).mean()


class _PAEDMulticlassFn(torch.autograd.Function):
    """softmax + one_hot + paed_loss_multiclass_soft and its gradient as libvitseg kernels (csrc/paed_loss.hip):
    one call yields the loss and d loss / d logits, the backward only scales it."""

    @staticmethod
    def forward(ctx, logits, target, sigma, class_penalty):
        from . import _lib
        logits = logits.contiguous().float()
        target = target.contiguous()
        if target.dtype not in (torch.int64, torch.uint8):
            raise ValueError(f"target must be int64 or uint8 class indices, got {target.dtype}")
        B, Cc, H, W = logits.shape
        if tuple(target.shape) != (B, H, W):
            raise ValueError(f"target shape {tuple(target.shape)} does not match logits {tuple(logits.shape)}")
        # (requires_grad of the cast copy says nothing inside Function.forward, where grad mode is off)
        need_grad = ctx.needs_input_grad[0]
        L = _lib.lib()
        scratch = torch.empty(L.vitseg_paed_scratch_bytes(B, Cc, H, W), dtype=torch.uint8, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        grad = torch.empty_like(logits) if need_grad else None
        with torch.cuda.device(logits.device):
            _lib.check(L.vitseg_paed_multiclass_loss(
                logits.data_ptr(), target.data_ptr(), int(target.dtype == torch.uint8), B, Cc, H, W, float(sigma),
                int(bool(class_penalty)), scratch.data_ptr(), loss.data_ptr(), None if grad is None else grad.data_ptr(),
                torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(*([grad] if grad is not None else []))
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.saved_tensors:
            raise RuntimeError("paed_multiclass_loss_fused: the gradient was not requested in the forward")
        return ctx.saved_tensors[0] * grad_out, None, None, None


def paed_multiclass_loss_fused(logits, target, sigma=3, class_penalty=True):
    """`paed_loss_multiclass_soft(one_hot(target), softmax(logits, 1), sigma=sigma, class_penalty=...)`
    (model/PAED/classes.py:336-369, called at :455-462) computed on the device in one fused call; differentiable w.r.t.
    `logits` [B, C, H, W].  `target`: class indices [B, H, W], int64 or uint8."""
    return _PAEDMulticlassFn.apply(logits, target, sigma, class_penalty)


class _PAEDBinaryFn(torch.autograd.Function):
    """sigmoid + BCE + 0.1 Dice + 5 |soft PAED| and d loss / d logits in three launches (csrc/paed_binary.hip)."""

    @staticmethod
    def forward(ctx, logits, mask, sdf_ext, sdf_int):
        from . import _lib
        logits = logits.contiguous().float()
        B, one, H, W = logits.shape
        if one != 1:
            raise ValueError(f"the binary PAED trainer expects [B, 1, H, W] logits, got {tuple(logits.shape)}")
        mask = mask.reshape(B, H, W).contiguous().float()
        sdf_ext, sdf_int = sdf_ext.contiguous().float(), sdf_int.contiguous().float()
        hs, ws = sdf_ext.shape[-2:]
        if tuple(sdf_int.shape[-2:]) != (hs, ws) or sdf_ext.numel() != B * hs * ws or sdf_int.numel() != B * hs * ws:
            raise ValueError("sdf_ext / sdf_int must be [B, h, w] maps of one size")
        L = _lib.lib()
        scratch = torch.empty(L.vitseg_paed_binary_scratch_bytes(B, H, W), dtype=torch.uint8, device=logits.device)
        out = torch.empty(8, dtype=torch.float32, device=logits.device)
        grad = torch.empty_like(logits) if ctx.needs_input_grad[0] else None   # (not logits.requires_grad: see above)
        with torch.cuda.device(logits.device):
            _lib.check(L.vitseg_paed_binary_loss(logits.data_ptr(), mask.data_ptr(), sdf_ext.data_ptr(), sdf_int.data_ptr(),
                                                 hs, ws, B, H, W, scratch.data_ptr(), out.data_ptr(),
                                                 None if grad is None else grad.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(*([grad] if grad is not None else []))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_loss, _grad_terms):
        if not ctx.saved_tensors:
            raise RuntimeError("paed_binary_loss_fused: the gradient was not requested in the forward")
        return ctx.saved_tensors[0] * grad_loss, None, None, None


def paed_binary_loss_fused(logits, mask, sdf_ext, sdf_int):
    """`F.binary_cross_entropy(p, mask) + 0.1 * dice_loss(p, mask) + 5 * |paed_loss_soft(sdf_ext, sdf_int, p)|` with
    p = sigmoid(logits) (model/PAED/classes.py:664-681), on the device in one fused call; differentiable w.r.t. `logits`
    [B, 1, H, W].  Returns (loss, terms) with terms = [loss, bce, dice, paed, tp, fp, fn, equal] (device floats)."""
    return _PAEDBinaryFn.apply(logits, mask, sdf_ext, sdf_int)


def dice_loss(preds, targets, smooth=1e-6):
    """1 - (2 |P.T| + s) / (|P| + |T| + s) on the flattened batch (model/PAED/classes.py:608-620)."""
    p, t = preds.float().reshape(-1), targets.float().reshape(-1)
    return 1 - (2.0 * (p * t).sum() + smooth) / (p.sum() + t.sum() + smooth)


def paed_loss_soft(gt_sdf_ext, gt_sdf_int, preds):
    """Soft PAED for the binary model (model/PAED/classes.py:623-661): the exterior SDF weights a
    max-normalised Sobel edge map of the prediction, the interior SDF rewards the prediction itself:
    1.0 * mean(sdf_ext * edge) - 0.5 * mean(sdf_int * pred).  SDFs [B,1,h,w] are bilinearly resized."""
    B, _, H, W = preds.shape
    ext = F.interpolate(gt_sdf_ext, size=(H, W), mode="bilinear", align_corners=False)
    inn = F.interpolate(gt_sdf_int, size=(H, W), mode="bilinear", align_corners=False)
    sx = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]], device=preds.device, dtype=preds.dtype).view(1, 1, 3, 3)
    gx = F.conv2d(preds, sx, padding=1)
    gy = F.conv2d(preds, sx.transpose(2, 3), padding=1)
    edge = torch.sqrt(gx ** 2 + gy ** 2 + 1e-6)
    edge = edge / (edge.view(B, -1).max(dim=1)[0].view(B, 1, 1, 1) + 1e-6)
    return 1 * (ext * edge).mean() - 0.5 * (inn * preds).mean()


_EVAL = {}


def iou_score(preds, targets, num_classes=17):
    """Mean over classes of the batch-mean IoU with 1e-6 smoothing (model/PAED/classes.py:430-447): per image and class
    inter = |pred = c and target = c|, union = |pred = c or target = c|.  On the device the per-image class statistics
    come from ONE launch (vitseg_eval_counts: |gt & pred|, |gt|, |pred| per label value) instead of the reference's loop of
    ~6 elementwise kernels per class; the rest is arithmetic on a [B, num_classes] table."""
    if preds.is_cuda and preds.dim() == 3 and preds.shape[1] == preds.shape[2] and num_classes <= 256:
        from .metrics import Evaluator
        key = (int(num_classes), str(preds.device))
        if key not in _EVAL:
            _EVAL[key] = Evaluator(num_classes, preds.device)
        c = _EVAL[key].counts(preds.to(torch.uint8), targets)[:, :, :num_classes].to(torch.float32)   # exact below 2^24 pixels
        inter, union = c[:, 0], c[:, 1] + c[:, 2] - c[:, 0]
        return ((inter + 1e-6) / (union + 1e-6)).mean(0).mean()
    ious = []
    for c in range(num_classes):
        p, t = (preds == c).float(), (targets == c).float()
        inter = (p * t).sum((1, 2))
        union = (p + t).clamp(0, 1).sum((1, 2))
        ious.append(((inter + 1e-6) / (union + 1e-6)).mean())
    return torch.stack(ious).mean()


class _Base(nn.Module):
    def __init__(self, num_classes, patch_size, hidden_size, num_hidden_layers, num_attention_heads, **kw):
        super().__init__()
        self.model = ViTSegmentationModel(num_classes, patch_size, hidden_size, num_hidden_layers,
                                          num_attention_heads, **kw)
        self.logged = {}

    def forward(self, x):
        return self.model(x)

    def _resize_target(self, y, size=None):
        size = size or (self.model.cfg.image_size,) * 2
        if y.dim() == 4:
            if y.shape[1] != 1:
                raise ValueError(f"Expected single-channel mask but got shape {y.shape}")  # :502-503
            y = y[:, 0]
        if y.is_cuda and y.dtype in (torch.long, torch.uint8):     # one gather kernel instead of unsqueeze/float/interpolate/long
            from .preprocess import Preprocessor
            if getattr(self, "_prep", None) is None or self._prep.device != y.device:
                self._prep = Preprocessor(self.model.cfg.image_size, device=y.device)
            return self._prep.targets(y, tuple(size))
        return F.interpolate(y.unsqueeze(1).float(), size=size, mode="nearest").squeeze(1).long()

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        return self.model.state_dict(destination=destination, prefix=prefix + "model.", keep_vars=keep_vars)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        return self.model.load_state_dict(state_dict, strict=strict)


class LightningViTModel(_Base):
    """model/PAED/classes.py:415-487 (the class count is forced to 17 there, :418)."""

    def __init__(self, num_classes, patch_size, hidden_size, num_hidden_layers, num_attention_heads, **kw):
        super().__init__(17, patch_size, hidden_size, num_hidden_layers, num_attention_heads, **kw)
        self.num_classes = 17

    def _step(self, batch, tag):
        x, y = batch
        y = self._resize_target(y)
        logits = self.forward(x)
        # softmax + one-hot + blur + weighting and their gradient run in libvitseg (csrc/paed_loss.hip)
        loss = paed_multiclass_loss_fused(logits, y)
        self.logged[f"{tag}_loss"] = loss.detach()                   # device scalars: no host sync per step
        self.logged[f"{tag}_iou"] = iou_score(logits.detach().argmax(dim=1), y, self.num_classes)  # argmax(softmax) = argmax
        return loss

    def training_step(self, batch, batch_idx):
        return self._step(batch, "train")

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            return self._step(batch, "valid")

    def configure_optimizers(self):
        from .optim import FusedAdam
        return FusedAdam(self.parameters(), lr=1e-4)  # :486-487


class PAEDTrainer(_Base):
    """model/PAED/classes.py:490-701; scripts build it with num_classes=1 (model/PAED/ViTscript.py:66)."""

    def _forward_step_paed(self, batch, batch_idx, tag="train"):
        images, masks, sdf_ext, sdf_int = batch
        masks = self._resize_target(masks).float()
        # sigmoid, BCE + 0.1 Dice + 5 |soft PAED| (:679-681), their gradient and the confusion counts: csrc/paed_binary.hip
        loss, terms = paed_binary_loss_fused(self.forward(images), masks, sdf_ext, sdf_int)
        with torch.no_grad():
            tp, fp, fn, eq = terms[4], terms[5], terms[6], terms[7]
            acc = eq / masks.numel()
            iou = tp / (tp + fp + fn).clamp_min(1e-12)
            dice = 2 * tp / (2 * tp + fp + fn).clamp_min(1e-12)
            prec, rec = tp / (tp + fp).clamp_min(1e-12), tp / (tp + fn).clamp_min(1e-12)
        # device scalars: nothing here forces a host sync per step (read them with float() when a value is wanted)
        self.logged.update({f"{tag}_loss": loss.detach(), f"{tag}_acc": acc, f"{tag}_IoU": iou, f"{tag}_dice": dice,
                            f"{tag}_precision": prec, f"{tag}_recall": rec})
        return loss, acc, iou, dice, prec, rec

    def training_step(self, batch, batch_idx):
        return self._forward_step_paed(batch, batch_idx, "train")[0]

    def validation_step(self, batch, batch_idx):
        with torch.no_grad():
            return self._forward_step_paed(batch, batch_idx, "val")[0]

    def configure_optimizers(self):
        from .optim import FusedAdamW
        opt = FusedAdamW(self.model.parameters(), lr=1e-4)  # AdamW(lr=1e-4), :536-548 -- one launch over the arena
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=30)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "monitor": "val_IoU", "interval": "epoch",
                                                   "frequency": 1}}
