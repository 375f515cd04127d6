"""ORACLE -- test infrastructure, NOT product code.

CPU restatement (pure torch, no `transformers`) of the reference hot path:
`ViTSegmentationModel.forward` + CE loss + Adam, as executed by
mtumalan/VisionTransformer's model/CE scripts.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (visiontransformer_amd/) never does.

Every function cites the reference lines it restates.  Paths are relative to
/root/reference unless they start with `transformers/` (= the third-party
HuggingFace source the reference calls into: transformers 5.15.0,
`transformers/models/vit/modeling_vit.py`; unpinned by the reference's
requirements.txt).

Parity pin: the reference holds no tests or golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against outputs of the REAL
reference class run in the build container -- `oracle/make_golden.py` extracts
`ViTSegmentationModel` from /root/reference/model/CE/classes.py, runs it on
transformers 5.15.0 / torch 2.10 CPU and commits the vectors under
`tests/golden/`; `tests/test_oracle_golden.py` checks this file against them.

All functions run in the dtype of their inputs (fp32 = the reference's
arithmetic; fp64 = a higher-precision truth for error budgeting).
"""
from __future__ import annotations

import math

import torch

HEAD_MID = 256


# ----------------------------------------------------------------------------- a2
def patch_embed(x, w, b, P):
    """Conv2d(3->D, kernel P, stride P) as a GEMM over patch rows.

    transformers/models/vit/modeling_vit.py:62-69 (`projection(x).flatten(2).transpose(1, 2)`).
    K order is (c, py, px); token t = gy * g + gx.
    """
    B, C, S, _ = x.shape
    g = S // P
    rows = x.reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    return rows @ w.reshape(w.shape[0], -1).T + b


# ----------------------------------------------------------------------------- a3
def embeddings(x, sd, P, drop=None):
    """cat(cls, patches) + position embeddings; eval mode (dropout off).

    transformers/models/vit/modeling_vit.py:129-161.
    """
    t = patch_embed(x, sd["backbone.embeddings.patch_embeddings.projection.weight"],
                    sd["backbone.embeddings.patch_embeddings.projection.bias"], P)
    cls = sd["backbone.embeddings.cls_token"].expand(t.shape[0], -1, -1)
    t = torch.cat([cls, t], dim=1) + sd["backbone.embeddings.position_embeddings"]
    if drop is not None:  # train mode: modeling_vit.py:159 `embeddings = self.dropout(embeddings)`
        t = t * drop.rows(0, 0, t.shape).to(t.dtype)
    return t


# ----------------------------------------------------------------------------- a4
def layer_norm(x, w, b, eps=1e-12):
    """nn.LayerNorm(D, eps=1e-12): biased variance over the last dim.

    transformers/models/vit/modeling_vit.py:261-262,348; configuration_vit.py:58.
    """
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


# ----------------------------------------------------------------------------- a5/a6
def attention(h, sd, prefix, A, drop=None, layer=0):
    """q/k/v Linear, softmax(q k^T * hd^-0.5) v, o_proj.  No mask, eval mode.

    transformers/models/vit/modeling_vit.py:164-189 (eager core; the default sdpa
    backend computes the same function), :207-238 (projections, head split).
    Returns (q, k, v, ctx, out) with q,k,v in [B, A, N, hd].
    """
    B, N, D = h.shape
    hd = D // A

    def proj(nm):
        return (h @ sd[prefix + f"attention.{nm}.weight"].T + sd[prefix + f"attention.{nm}.bias"]) \
            .reshape(B, N, A, hd).transpose(1, 2)

    q, k, v = proj("q_proj"), proj("k_proj"), proj("v_proj")
    s = torch.softmax((q @ k.transpose(-1, -2)) * (hd ** -0.5), dim=-1)
    if drop is not None:  # modeling_vit.py:184 dropout on the attention probabilities
        s = s * drop.attn(layer, s.shape).to(s.dtype)
    ctx = (s @ v).transpose(1, 2).reshape(B, N, D)
    out = ctx @ sd[prefix + "attention.o_proj.weight"].T + sd[prefix + "attention.o_proj.bias"]
    if drop is not None:  # modeling_vit.py:276 hidden dropout after the attention output projection
        out = out * drop.rows(layer, 2, out.shape).to(out.dtype)
    return q, k, v, ctx, out


# ----------------------------------------------------------------------------- a7
def gelu_erf(u):
    """Exact (erf) GELU: configuration_vit.py:54 `hidden_act="gelu"`, activations.py:78-83."""
    return 0.5 * u * (1.0 + torch.erf(u / math.sqrt(2.0)))


def mlp(h, sd, prefix, drop=None, layer=0):
    """fc2(gelu(fc1(h))).  transformers/models/vit/modeling_vit.py:249-254 (+ hidden dropout, :283)."""
    u = gelu_erf(h @ sd[prefix + "mlp.fc1.weight"].T + sd[prefix + "mlp.fc1.bias"])
    out = u @ sd[prefix + "mlp.fc2.weight"].T + sd[prefix + "mlp.fc2.bias"]
    if drop is not None:
        out = out * drop.rows(layer, 3, out.shape).to(out.dtype)
    return out


# ----------------------------------------------------------------------------- a8
def encoder_layer(t, sd, i, A, eps=1e-12, stages=None, drop=None):
    """Pre-LN block.  transformers/models/vit/modeling_vit.py:266-286."""
    p = f"backbone.layers.{i}."
    h = layer_norm(t, sd[p + "layernorm_before.weight"], sd[p + "layernorm_before.bias"], eps)
    q, k, v, ctx, a = attention(h, sd, p, A, drop, i)
    t = t + a
    h2 = layer_norm(t, sd[p + "layernorm_after.weight"], sd[p + "layernorm_after.bias"], eps)
    m = mlp(h2, sd, p, drop, i)
    if stages is not None and i == 0:
        stages.update(ln1_0=h, q_0=q, k_0=k, v_0=v, ctx_0=ctx, attn_res_0=t, mlp_0=m)
    return t + m


# ----------------------------------------------------------------------------- a9
def encoder(x, sd, cfg, stages=None, drop=None):
    """ViTModel.forward minus the discarded pooler.  modeling_vit.py:356-388.
    `drop` (optional, train mode): an object with rows(layer, site, shape) / attn(layer, shape) returning the
    multiplicative dropout masks (keep -> 1/(1-p), drop -> 0) so that a test can inject the build's masks."""
    t = embeddings(x, sd, cfg.patch_size, drop)
    if stages is not None:
        stages["embeddings"] = t
    for i in range(cfg.num_hidden_layers):
        t = encoder_layer(t, sd, i, cfg.num_attention_heads, cfg.layer_norm_eps, stages, drop)
        if stages is not None:
            stages[f"layer_{i}"] = t
    t = layer_norm(t, sd["backbone.layernorm.weight"], sd["backbone.layernorm.bias"], cfg.layer_norm_eps)
    if stages is not None:
        stages["last_hidden_state"] = t
    return t


# ----------------------------------------------------------------------------- a10/a11
def seg_head(hidden, sd, stages=None):
    """Drop CLS, tokens -> NCHW, Conv3x3(pad 1) + ReLU + Conv1x1.

    model/CE/classes.py:250-257 (glue + seg_head), :240-244 (definition).  stages["head_pre"] = the ReLU's input
    (the tests use it to name the units whose sign an fp32 evaluation can flip).
    """
    B, N, D = hidden.shape
    g = int((N - 1) ** 0.5)
    f = hidden[:, 1:, :].transpose(1, 2).reshape(B, D, g, g)
    pre = torch.nn.functional.conv2d(f, sd["seg_head.0.weight"], sd["seg_head.0.bias"], padding=1)
    if stages is not None:
        stages["head_pre"] = pre
    return torch.nn.functional.conv2d(torch.relu(pre), sd["seg_head.2.weight"], sd["seg_head.2.bias"])


# ----------------------------------------------------------------------------- a12
def bilinear_taps(n_in, n_out, dtype):
    """Source taps of F.interpolate(mode='bilinear', align_corners=False).

    model/CE/classes.py:260.  src = max((d + 0.5) * in/out - 0.5, 0); i0 = floor(src);
    i1 = min(i0 + 1, in - 1); lam = src - i0.  The coordinate is computed in the
    tensor's arithmetic type (ATen area_pixel_compute_source_index).
    """
    scale = torch.tensor(n_in / n_out, dtype=dtype)
    d = torch.arange(n_out, dtype=dtype)
    src = torch.clamp((d + 0.5) * scale - 0.5, min=0.0)
    i0 = src.floor().long()
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    lam = src - i0.to(dtype)
    return i0, i1, lam


def _fma(a, b, c):
    """round(a*b + c) with a single rounding.  fp32 operands: the product is exact in fp64
    and the sum is rounded once more to fp32 (double rounding only on exact fp32 midpoints);
    fp64 operands: plain arithmetic (used only as a higher-precision truth)."""
    if a.dtype == torch.float32:
        return (a.double() * b.double() + c.double()).float()
    return a * b + c


def upsample_bilinear(z, size):
    """Separable bilinear upsample in the exact arithmetic order of ATen's CPU kernel
    (UpSampleKernel.cpp `Interpolate<n>::eval`: `out = t0*w0; out += t1*w1`, W inside H).
    As compiled for x86 with FMA the accumulation contracts to
        row = fma(a, wx0, b*wx1);   out = fma(row_top, wy0, row_bot*wy1)
    (established bit-for-bit against F.interpolate in oracle/make_golden.py's container:
    0 mismatches over all golden cases; every other fma/no-fma placement mismatches >20%).
    Domain: output H + W > 128.  Below that ATen dispatches to a different (vectorised) kernel
    (`_use_vectorized_kernel_cond_2d`, UpSampleKernel.cpp) whose rounding differs in the last
    bit; every size the reference uses (224, and BASELINE's 512 / 1024) is in the domain."""
    H, W = size
    y0, y1, ly = bilinear_taps(z.shape[2], H, z.dtype)
    x0, x1, lx = bilinear_taps(z.shape[3], W, z.dtype)
    top = z[:, :, y0, :]
    bot = z[:, :, y1, :]
    wx0, wx1 = (1.0 - lx).expand(top.shape[:-1] + (W,)), lx
    t = _fma(top[..., x0], wx0, top[..., x1] * wx1)
    b = _fma(bot[..., x0], wx0, bot[..., x1] * wx1)
    wy0, wy1 = (1.0 - ly)[:, None].expand_as(t), ly[:, None]
    return _fma(t, wy0, b * wy1)


# ----------------------------------------------------------------------------- a1..a12
def forward(x, sd, cfg, stages=None, drop=None):
    """ViTSegmentationModel.forward, eval mode.  model/CE/classes.py:246-262.

    Raises like the reference: ValueError on channel / image-size mismatch
    (transformers/models/vit/modeling_vit.py:63-68, :152-156).
    """
    if x.shape[1] != cfg.num_channels:
        raise ValueError("Make sure that the channel dimension of the pixel values match with the one set in the "
                         f"configuration. Expected {cfg.num_channels} but got {x.shape[1]}.")
    if x.shape[2] != cfg.image_size or x.shape[3] != cfg.image_size:
        raise ValueError(f"Input image size ({x.shape[2]}*{x.shape[3]}) doesn't match model "
                         f"({cfg.image_size}*{cfg.image_size}).")
    hidden = encoder(x, sd, cfg, stages, drop)
    z = seg_head(hidden, sd, stages)
    if stages is not None:
        stages["lowres_logits"] = z
    return upsample_bilinear(z, x.shape[2:])


# ----------------------------------------------------------------------------- a14
def _exp_sleef_u10(d):
    """fp32 `Sleef_expf{8,16}_u10` (the `Vectorized<float>::exp()` of ATen's AVX2 / AVX-512 builds), restated
    operation by operation: q = rint(d / ln 2); s = d - q ln2 (two-constant Cody-Waite, fused); degree-6
    polynomial in s by fused multiply-adds; 1 + (s + s^2 u); scaled by 2^(q>>1) and 2^(q - (q>>1)).
    Third-party arithmetic (Sleef 3.x `xexpf`, vendored by torch 2.10): pinned bit-for-bit against
    torch.sigmoid in tests/test_oracle_golden.py::test_sigmoid_restatement_is_atens."""
    f = torch.float32
    q = torch.round(d * torch.tensor(1.4426950408889634, dtype=f))      # rint: ties to even
    s = _fma(q, torch.tensor(-0.693145751953125, dtype=f), d)
    s = _fma(q, torch.tensor(-1.428606765330187045e-06, dtype=f), s)
    u = torch.full_like(d, 0.000198527617612853646278381)
    for c in (0.00139304355252534151077271, 0.00833336077630519866943359, 0.0416664853692054748535156,
              0.166666671633720397949219, 0.5):
        u = _fma(u, s, torch.tensor(c, dtype=f))
    u = 1.0 + _fma(s * s, u, s)
    qi = q.to(torch.int32)
    h = qi >> 1

    def pow2(e):
        return ((e + 127) << 23).view(torch.float32)

    u = (u * pow2(h)) * pow2(qi - h)
    u = torch.where(d < -104.0, torch.zeros_like(u), u)
    return torch.where(d > 100.0, torch.full_like(u, float("inf")), u)


def sigmoid_aten(x):
    """`logits.sigmoid()` as ATen's CPU kernel computes it for fp32 (UnaryOpsKernel.cpp sigmoid_kernel, vector
    path): a = 0 - x; a = a.exp(); a = 1 + a; a = 1 / a.  Restated so that the mask decision -- which hinges on
    fp32 sigmoid TIES between classes -- does not depend on the host's torch build or thread partition (the
    scalar tail of a parallel chunk goes through glibc's expf instead).  fp64 inputs: plain sigmoid."""
    if x.dtype != torch.float32:
        return x.sigmoid()
    return 1.0 / (1.0 + _exp_sleef_u10(0.0 - x))


def predict_mask(logits):
    """Inference post-processing of the reference scripts: sigmoid THEN argmax over
    classes, first maximal index wins.  model/CE/testViTModel.py:122-126,
    model/CE/datasetTestViTmodel.py:183-190."""
    return sigmoid_aten(logits).argmax(dim=1)


def mask_stable(logits, eps):
    """Pixels whose sigmoid -> first-max decision cannot change when every logit moves by at most `eps`
    (used by the parity tests to exempt ONLY the pixels a logit error of that size can flip): the winner w must
    stay strictly above every earlier class and at least level with every later one."""
    w = predict_mask(logits)
    lo = sigmoid_aten(logits.gather(1, w[:, None]) - eps)      # the winner, pushed down
    hi = sigmoid_aten(logits + eps)                            # everybody else, pushed up
    cls = torch.arange(logits.shape[1]).view(1, -1, 1, 1)
    ok = torch.where(cls < w[:, None], hi < lo, hi <= lo) | (cls == w[:, None])
    return ok.all(dim=1)


# ----------------------------------------------------------------------------- a13
def resize_target(y, size):
    """F.interpolate(y[:,None].float(), size, mode='nearest').long(): idx = min(floor(dst*in/out), in-1).

    model/CE/classes.py:273-274.  ATen computes the source index in fp32:
    floor(dst * float(in)/float(out)).
    """
    H, W = size

    def idx(n_in, n_out):
        scale = torch.tensor(n_in / n_out, dtype=torch.float32)
        return torch.clamp((torch.arange(n_out, dtype=torch.float32) * scale).floor().long(), max=n_in - 1)

    return y[:, idx(y.shape[1], H)][:, :, idx(y.shape[2], W)]


def ce_loss(logits, target):
    """nn.CrossEntropyLoss() defaults: mean over B*H*W of -log_softmax(logits)[target].

    model/CE/classes.py:268,280.
    """
    lse = torch.logsumexp(logits, dim=1)
    picked = logits.gather(1, target[:, None]).squeeze(1)
    return (lse - picked).mean()


def adam_step(p, g, m, v, step, lr=1e-5, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(lr=1e-5) single-tensor update (model/CE/classes.py:296-297),
    defaults betas (0.9, 0.999), eps 1e-8, weight_decay 0, amsgrad False.  `step` is 1-based.
    Returns (p, m, v) new values."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


def training_step(x, y, sd, cfg, target_size=None):
    """LightningViTModel.training_step minus logging (model/CE/classes.py:276-285):
    nearest-resize targets, forward (dropout p=0 for parity), CE.
    The reference hard-codes size=(224, 224) (= its image_size); `target_size`
    defaults to the model's image size.  Returns (loss, grads dict) via autograd
    on this restatement."""
    S = cfg.image_size if target_size is None else target_size
    y = resize_target(y, (S, S))
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = ce_loss(forward(x, leaf, cfg), y)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaf.items()}
