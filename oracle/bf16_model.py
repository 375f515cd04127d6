"""ORACLE -- test infrastructure, NOT product code.

The ROUNDING MODEL of libvitseg's bf16 mixed-precision training step, in fp64: the arithmetic of
`vitseg_oracle.py` (= the reference's model/CE training step, /root/reference/model/CE/classes.py:276-285 over
transformers/models/vit/modeling_vit.py:164-286) with every value rounded to bf16 exactly where the HIP path
(visiontransformer_amd/csrc/vitseg_train.hip forward_train_bf16 / backward_bf16) stores or multiplies a bf16:

  forward   GEMM operands (LayerNorm outputs, the 16-bit shadow of the weights, q|k|v, the attention context,
            gelu(u)) are bf16; accumulation, bias, residual stream, LayerNorm statistics, softmax are wider (fp32 on
            the GPU, fp64 here); the attention probabilities are rounded before P.V; gelu'(u) is SAVED as bf16;
  backward  every gradient that travels between kernels as a GEMM operand is bf16: the gradient entering a residual
            branch (after its dropout mask), dU (after the multiplication by the saved gelu'), dH, dCTX, dQ|dK|dV;
            inside the attention backward P~ (for dV) and dS (for dQ, dK) are rounded, and delta = sum_d dO O uses the
            bf16 context the forward wrote; parameter gradients and the residual-stream gradient stay wide; the head's
            dF is rounded for its two GEMMs; the patch-embedding weight gradient multiplies bf16(dX) by bf16 patches.

What it is for: the distance of the HIP bf16 gradients from the exact (fp64) gradients is set by these roundings, not by
the kernels -- IF the kernels implement this model.  tests/test_gpu_backward.py therefore (1) holds each bf16 attention
kernel to this model on identical inputs (one rounding stage deep: the realisations coincide, the gap is ~1e-3 even where
the gap to exact arithmetic is 0.2), and (2) derives the full-depth gradient gates from the model's own distance to the
exact gradients instead of from a constant fitted to a measured run.  (Through 12 layers the two realisations of the
rounding noise decorrelate -- a flipped bf16 rounding perturbs every later rounding decision -- so at depth the
comparison is of error LEVELS per tensor, not element by element.)

Not modelled (each below 1e-4 of the gradient): fp32 instead of fp64 accumulation; the forward rounding P relative to
the running maximum instead of the normalised P; fc1's bias gradient summed before dU is rounded; the fp32x3 patch GEMM.
Only tests import this module.
"""
from __future__ import annotations

import math

import torch

from . import vitseg_oracle as O


def rb(t: torch.Tensor) -> torch.Tensor:
    """round to nearest-even bf16, returned in the input's dtype"""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundBoth(torch.autograd.Function):
    """a bf16 tensor in the forward whose gradient is a bf16 tensor in the backward"""

    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return rb(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return rb(g)


class _GeluSaved(torch.autograd.Function):
    """fc1's epilogue: writes bf16 gelu(u) and bf16 gelu'(u); the backward GEMM's epilogue multiplies by the saved
    derivative and rounds (gemm_p8.hip EPI_GELU / EPI_DGELU)."""

    @staticmethod
    def forward(ctx, u):
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        pdf = torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        ctx.save_for_backward(rb(cdf + u * pdf))
        return rb(u * cdf)

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return rb(g * d)


class _AttnCore(torch.autograd.Function):
    """softmax(q k^T hd^-1/2) (dropout) v on bf16 q, k, v [B, A, N, hd] as attention_bf16.hip / attention_bwd_bf16.hip
    compute it (header of attention_bwd_bf16.hip).  `mask`: keep / (1 - p) multipliers [B, A, N, N] or None."""

    @staticmethod
    def forward(ctx, q, k, v, mask):
        hd = q.shape[-1]
        s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
        lse = torch.logsumexp(s, dim=-1, keepdim=True)
        p = torch.exp(s - lse)
        pd = p if mask is None else p * mask
        o = rb(rb(pd) @ v)                 # the context leaves the kernel as bf16
        ctx.save_for_backward(q, k, v, lse, o)
        ctx.mask = mask
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, lse, o = ctx.saved_tensors
        return (*attention_backward(q, k, v, do, o, lse, ctx.mask), None)


def attention_backward(q, k, v, do, o, lse, mask=None):
    """(dq, dk, dv), unrounded, from bf16 q, k, v [B, A, N, hd], the incoming gradient `do`, the bf16 context `o` the
    forward wrote and its natural-log log-sum-exp `lse` [B, A, N, 1]: the arithmetic of attention_bwd_bf16.hip."""
    hd = q.shape[-1]
    do = rb(do)                        # dCTX is written as bf16 by the o_proj dgrad
    p = torch.exp((q @ k.transpose(-1, -2)) * (hd ** -0.5) - lse)
    delta = (do * o).sum(dim=-1, keepdim=True)
    dpd = do @ v.transpose(-1, -2)
    dp = dpd if mask is None else dpd * mask
    ds = rb(p * (dp - delta))
    pdb = rb(p if mask is None else p * mask)
    dv = pdb.transpose(-1, -2) @ do
    dq = (ds @ k) * (hd ** -0.5)
    dk = (ds.transpose(-1, -2) @ q) * (hd ** -0.5)
    return dq, dk, dv


def attention_core(q, k, v, mask=None):
    return _AttnCore.apply(q, k, v, mask)


class _PatchEmbed(torch.autograd.Function):
    """forward: the fp32-grade patch GEMM; backward: dW = bf16(dX)^T bf16(patch rows) (vitseg_train.hip section 5)."""

    @staticmethod
    def forward(ctx, rows, w2d, b):
        ctx.save_for_backward(rows)
        return rows @ w2d.T + b

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        g2, r2 = g.reshape(-1, g.shape[-1]), rows.reshape(-1, rows.shape[-1])
        return None, rb(g2).T @ rb(r2), g2.sum(dim=0)


def _linear(h, w, b):
    """bf16 activations x the bf16 shadow of the fp32 master weight, wide accumulate, fp32 bias"""
    return h @ _RoundFwd.apply(w).T + b


def forward(x, sd, cfg, drop=None):
    """logits of the bf16 training forward (dropout masks injected through `drop` as in vitseg_oracle.encoder)."""
    P, A, eps = cfg.patch_size, cfg.num_attention_heads, cfg.layer_norm_eps
    B, C, S, _ = x.shape
    g = S // P
    rows = x.reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    w = sd["backbone.embeddings.patch_embeddings.projection.weight"]
    t = _PatchEmbed.apply(rows, w.reshape(w.shape[0], -1), sd["backbone.embeddings.patch_embeddings.projection.bias"])
    cls = sd["backbone.embeddings.cls_token"].expand(B, -1, -1)
    t = torch.cat([cls, t], dim=1) + sd["backbone.embeddings.position_embeddings"]
    if drop is not None:
        t = t * drop.rows(0, 0, t.shape).to(t.dtype)
    N, D = t.shape[1], t.shape[2]
    hd = D // A
    for i in range(cfg.num_hidden_layers):
        p = f"backbone.layers.{i}."
        h = _RoundBoth.apply(O.layer_norm(t, sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], eps))
        wqkv = torch.cat([sd[p + f"attention.{n}.weight"] for n in ("q_proj", "k_proj", "v_proj")], dim=0)
        bqkv = torch.cat([sd[p + f"attention.{n}.bias"] for n in ("q_proj", "k_proj", "v_proj")], dim=0)
        qkv = _RoundBoth.apply(_linear(h, wqkv, bqkv))
        q, k, v = [qkv[..., j * D:(j + 1) * D].reshape(B, N, A, hd).transpose(1, 2) for j in range(3)]
        mask = drop.attn(i, (B, A, N, N)).to(t.dtype) if drop is not None else None
        ctx = attention_core(q, k, v, mask).transpose(1, 2).reshape(B, N, D)
        a = _RoundGrad.apply(_linear(ctx, sd[p + "attention.o_proj.weight"], sd[p + "attention.o_proj.bias"]))
        if drop is not None:
            a = a * drop.rows(i, 2, a.shape).to(a.dtype)
        t = t + a
        h2 = _RoundBoth.apply(O.layer_norm(t, sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], eps))
        u = _GeluSaved.apply(_linear(h2, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        m = _RoundGrad.apply(_linear(u, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]))
        if drop is not None:
            m = m * drop.rows(i, 3, m.shape).to(m.dtype)
        t = t + m
    hf = _RoundFwd.apply(O.layer_norm(t, sd["backbone.layernorm.weight"], sd["backbone.layernorm.bias"], eps))
    f = hf[:, 1:, :].transpose(1, 2).reshape(B, D, g, g)
    pre = _RoundGrad.apply(torch.nn.functional.conv2d(f, _RoundFwd.apply(sd["seg_head.0.weight"]), None, padding=1))
    pre = pre + sd["seg_head.0.bias"].view(1, -1, 1, 1)
    z = torch.nn.functional.conv2d(torch.relu(pre), sd["seg_head.2.weight"], sd["seg_head.2.bias"])
    return O.upsample_bilinear(z, x.shape[2:])


def training_step(x, y, sd, cfg, drop=None):
    """(loss, {name: gradient}) of the rounding model: fp64 arithmetic, bf16 roundings as listed in the module header."""
    leaf = {k: v.detach().double().clone().requires_grad_(True) for k, v in sd.items()}
    loss = O.ce_loss(forward(x.double(), leaf, cfg, drop), y)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaf.items()}
