"""Parameter arena <-> reference state-dict schema.

The C library owns the arena layout (vitseg_param_offset); this module maps the reference's
parameter names onto views of that arena so `state_dict()` / `load_state_dict()` keep the
reference's key schema:
  * transformers 5.x names (what the container's reference produces),
  * transformers 4.x legacy names (the era of the reference's private checkpoints;
    transformers/conversion_mapping.py:338-346),
  * an optional Lightning `model.` prefix (model/CE/classes.py:267, testViTModel.py:118),
  * `backbone.pooler.dense.*` tolerated and ignored (computed-then-discarded in the reference).
"""
from __future__ import annotations

import re
from typing import Dict

import torch

from . import _lib
from .config import ViTSegConfig, HEAD_MID_CHANNELS

_LEGACY = [
    (r"backbone\.encoder\.layer\.(\d+)\.attention\.attention\.query\.", r"backbone.layers.\1.attention.q_proj."),
    (r"backbone\.encoder\.layer\.(\d+)\.attention\.attention\.key\.", r"backbone.layers.\1.attention.k_proj."),
    (r"backbone\.encoder\.layer\.(\d+)\.attention\.attention\.value\.", r"backbone.layers.\1.attention.v_proj."),
    (r"backbone\.encoder\.layer\.(\d+)\.attention\.output\.dense\.", r"backbone.layers.\1.attention.o_proj."),
    (r"backbone\.encoder\.layer\.(\d+)\.intermediate\.dense\.", r"backbone.layers.\1.mlp.fc1."),
    (r"backbone\.encoder\.layer\.(\d+)\.output\.dense\.", r"backbone.layers.\1.mlp.fc2."),
    (r"backbone\.encoder\.layer\.(\d+)\.(layernorm_before|layernorm_after)\.", r"backbone.layers.\1.\2."),
]


def canonical_key(key: str) -> str:
    """Any accepted spelling -> transformers-5.x name without the Lightning prefix."""
    if key.startswith("model."):
        key = key[len("model."):]
    for pat, rep in _LEGACY:
        key, n = re.subn("^" + pat, rep, key)
        if n:
            break
    return key


def arena_views(cfg: ViTSegConfig, arena: torch.Tensor) -> Dict[str, torch.Tensor]:
    """reference name -> view of `arena` with the reference's shape (no copies).

    seg_head.0.weight is stored (out, ky, kx, in) for the implicit-GEMM kernel; its view is the
    permuted (out, in, ky, kx) the reference uses, so it is non-contiguous.
    """
    D, I, P, C = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size, cfg.num_classes

    def t(tid, layer=0):
        off, n = _lib.param_offset(cfg, tid, layer)
        return arena[off:off + n]

    v = {
        "backbone.embeddings.cls_token": t(_lib.T_CLS).view(1, 1, D),
        "backbone.embeddings.position_embeddings": t(_lib.T_POS).view(1, cfg.seq_len, D),
        "backbone.embeddings.patch_embeddings.projection.weight": t(_lib.T_PATCH_W).view(D, cfg.num_channels, P, P),
        "backbone.embeddings.patch_embeddings.projection.bias": t(_lib.T_PATCH_B),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"backbone.layers.{i}."
        wqkv = t(_lib.T_WQKV, i).view(3, D, D)
        bqkv = t(_lib.T_BQKV, i).view(3, D)
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
            v[p + f"attention.{nm}.weight"] = wqkv[j]
            v[p + f"attention.{nm}.bias"] = bqkv[j]
        v[p + "attention.o_proj.weight"] = t(_lib.T_WO, i).view(D, D)
        v[p + "attention.o_proj.bias"] = t(_lib.T_BO, i)
        v[p + "layernorm_before.weight"] = t(_lib.T_LN1_W, i)
        v[p + "layernorm_before.bias"] = t(_lib.T_LN1_B, i)
        v[p + "layernorm_after.weight"] = t(_lib.T_LN2_W, i)
        v[p + "layernorm_after.bias"] = t(_lib.T_LN2_B, i)
        v[p + "mlp.fc1.weight"] = t(_lib.T_W1, i).view(I, D)
        v[p + "mlp.fc1.bias"] = t(_lib.T_B1, i)
        v[p + "mlp.fc2.weight"] = t(_lib.T_W2, i).view(D, I)
        v[p + "mlp.fc2.bias"] = t(_lib.T_B2, i)
    v["backbone.layernorm.weight"] = t(_lib.T_LNF_W)
    v["backbone.layernorm.bias"] = t(_lib.T_LNF_B)
    v["seg_head.0.weight"] = t(_lib.T_HEAD0_W).view(HEAD_MID_CHANNELS, 3, 3, D).permute(0, 3, 1, 2)
    v["seg_head.0.bias"] = t(_lib.T_HEAD0_B)
    v["seg_head.2.weight"] = t(_lib.T_HEAD2_W).view(C, HEAD_MID_CHANNELS, 1, 1)
    v["seg_head.2.bias"] = t(_lib.T_HEAD2_B)
    return v
