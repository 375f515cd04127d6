"""Procedural (counter-based) synthetic weights, images and targets.

There is no network for checkpoints or datasets, and the reference's own weights
are private, so every test, golden fixture and benchmark uses tensors generated
here.  The generator is a pure function  value = f(seed, tensor-name, index)
built from integer hashing plus IEEE add/multiply only (no libm), so the GPU
box regenerates bit-identical tensors from the committed code alone.

Distributions follow the reference's initialisers:
backbone Linear/Conv weights ~ N(0, 0.02)-like (Irwin-Hall of 4 uniforms, unit
variance), biases 0, LayerNorm 1/0, cls/pos ~ N(0, 0.02)-like; `seg_head`
Kaiming-uniform U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's Conv2d default.
`perturb=True` additionally jitters biases and LayerNorm affine parameters so
that parity tests exercise them (an all-zero bias hides a missing bias add).
"""
from __future__ import annotations

import numpy as np

from .config import ViTSegConfig, HEAD_MID_CHANNELS

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_SQRT3 = 1.7320508075688772


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser, vectorised over uint64 arrays (wraps mod 2^64)."""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(seed: int, name: str, n: int, draw: int = 0) -> np.ndarray:
    """n float64 values in [0, 1): element i = hash(seed, name, draw, i) >> 11 * 2^-53."""
    with np.errstate(over="ignore"):
        key = np.uint64((_fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
                         ^ ((draw * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF)
        idx = np.arange(n, dtype=np.uint64)
        h = _mix(_mix(idx * np.uint64(0x9E3779B97F4A7C15) + key) ^ key)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal_like(seed: int, name: str, n: int) -> np.ndarray:
    """Unit-variance, zero-mean, bounded (|z| <= 2*sqrt(3)) bell-shaped variate."""
    s = uniform01(seed, name, n, 0)
    for d in (1, 2, 3):
        s = s + uniform01(seed, name, n, d)
    return (s - 2.0) * _SQRT3


def param_shapes(cfg: ViTSegConfig) -> dict:
    """Reference state-dict schema (transformers 5.x names), SURVEY.md section 8 a1.

    The unused pooler (`modeling_vit.py:386`) is not part of the hot path and is omitted.
    """
    D, I, P, C = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_classes
    sh = {
        "backbone.embeddings.cls_token": (1, 1, D),
        "backbone.embeddings.position_embeddings": (1, cfg.seq_len, D),
        "backbone.embeddings.patch_embeddings.projection.weight": (D, cfg.num_channels, P, P),
        "backbone.embeddings.patch_embeddings.projection.bias": (D,),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"backbone.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sh[p + f"attention.{nm}.weight"] = (D, D)
            sh[p + f"attention.{nm}.bias"] = (D,)
        for nm in ("layernorm_before", "layernorm_after"):
            sh[p + nm + ".weight"] = (D,)
            sh[p + nm + ".bias"] = (D,)
        sh[p + "mlp.fc1.weight"] = (I, D)
        sh[p + "mlp.fc1.bias"] = (I,)
        sh[p + "mlp.fc2.weight"] = (D, I)
        sh[p + "mlp.fc2.bias"] = (D,)
    sh["backbone.layernorm.weight"] = (D,)
    sh["backbone.layernorm.bias"] = (D,)
    sh["seg_head.0.weight"] = (HEAD_MID_CHANNELS, D, 3, 3)
    sh["seg_head.0.bias"] = (HEAD_MID_CHANNELS,)
    sh["seg_head.2.weight"] = (C, HEAD_MID_CHANNELS, 1, 1)
    sh["seg_head.2.bias"] = (C,)
    return sh


def make_state_dict(cfg: ViTSegConfig, seed: int = 1, perturb: bool = True, head_gain: float = 1.0) -> dict:
    """name -> float32 numpy array for every hot-path parameter.

    `head_gain` scales seg_head.2 weights/bias (used to drive logits into the
    fp32-sigmoid saturation regime for the sigmoid-then-argmax tie tests).
    """
    out = {}
    for name, shape in param_shapes(cfg).items():
        n = int(np.prod(shape))
        if name.startswith("seg_head."):
            w_shape = param_shapes(cfg)[name.rsplit(".", 1)[0] + ".weight"]
            fan_in = int(np.prod(w_shape[1:]))
            bound = 1.0 / (fan_in ** 0.5)
            v = (uniform01(seed, name, n) * 2.0 - 1.0) * bound
            if name.startswith("seg_head.2."):
                v = v * head_gain
        elif "layernorm" in name:
            base = 1.0 if name.endswith(".weight") else 0.0
            v = np.full(n, base) + (0.1 * normal_like(seed, name, n) if perturb else 0.0)
        elif name.endswith(".bias"):
            v = 0.02 * normal_like(seed, name, n) if perturb else np.zeros(n)
        else:
            v = 0.02 * normal_like(seed, name, n)
        out[name] = np.ascontiguousarray(v.reshape(shape).astype(np.float32))
    return out


def make_images(cfg: ViTSegConfig, batch: int, seed: int = 0, first_image: int = 0) -> np.ndarray:
    """float32 NCHW in [0,1), like `Resize -> ToTensor` output (trainCurrentViTmodel.py:48-51).

    Image i depends only on (seed, first_image + i) so data-parallel ranks can
    generate disjoint shards of one global batch.
    """
    S, Cin = cfg.image_size, cfg.num_channels
    imgs = [uniform01(seed, f"image.{first_image + i}", Cin * S * S).reshape(Cin, S, S) for i in range(batch)]
    return np.stack(imgs).astype(np.float32)


def make_targets(cfg: ViTSegConfig, batch: int, seed: int = 0, first_image: int = 0, size: int = 256) -> np.ndarray:
    """int64 class indices [B, size, size] in [0, C): blocky (8x8 cells) like real masks
    (dataset masks are 256x256 nearest-resized, classes.py:76-89)."""
    cell = 8
    gs = size // cell
    out = []
    for i in range(batch):
        u = uniform01(seed, f"target.{first_image + i}", gs * gs).reshape(gs, gs)
        t = np.minimum((u * cfg.num_classes).astype(np.int64), cfg.num_classes - 1)
        out.append(np.repeat(np.repeat(t, cell, axis=0), cell, axis=1))
    return np.stack(out)
