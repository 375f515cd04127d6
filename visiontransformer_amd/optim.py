"""Fused Adam over the flat parameter arena (one kernel launch per step).

Same update as `torch.optim.Adam(params, lr)` with default betas/eps, weight_decay 0, amsgrad off --
what `LightningViTModel.configure_optimizers` builds in the reference (model/CE/classes.py:296-297).
"""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        """`weight_decay` is DECOUPLED (torch.optim.AdamW's: p *= 1 - lr * weight_decay in front of the update); 0 = Adam."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.numel() % 4 == 0):
                    raise RuntimeError("FusedAdam needs contiguous fp32 HIP tensors whose size is a multiple of 4")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                g = p.grad.contiguous()
                with torch.cuda.device(p.device):
                    _lib.check(_lib.lib().vitseg_adamw_step(
                        p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                        group["lr"], group["betas"][0], group["betas"][1], group["eps"], group.get("weight_decay", 0.0),
                        st["step"], grad_scale, torch.cuda.current_stream().cuda_stream))
                # the update went through the raw pointer: bump the version counter (no kernel) so the bf16 shadow of
                # the arena and autograd's saved-tensor checks see the modification
                torch.autograd.graph.increment_version(p)
        return loss


class FusedAdamW(FusedAdam):
    """`torch.optim.AdamW(params, lr)` (default weight_decay 1e-2) over the flat arena in one launch per step: what
    `PAEDTrainer.configure_optimizers` builds in the reference (model/PAED/classes.py:536-548)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
