"""ctypes binding of csrc/libvitseg.so (C ABI: include/vitseg.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing
this module's `lib()` raises -- the product path must never silently run elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os

from .config import ViTSegConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvitseg.so")

OK, EINVAL, ESHAPE, EWORKSPACE, EHIP = 0, -1, -2, -3, -4
F32, BF16, F16, F32X3 = 0, 1, 2, 3
BUF_TOKENS, BUF_LOWRES = 0, 1

# enum vitseg_tensor
(T_CLS, T_POS, T_PATCH_W, T_PATCH_B, T_LN1_W, T_LN1_B, T_WQKV, T_BQKV, T_WO, T_BO, T_LN2_W, T_LN2_B,
 T_W1, T_B1, T_W2, T_B2, T_LNF_W, T_LNF_B, T_HEAD0_W, T_HEAD0_B, T_HEAD2_W, T_HEAD2_B, T_COUNT) = range(23)

EXPORTS = [
    "vitseg_version", "vitseg_last_error", "vitseg_set_option", "vitseg_get_option", "vitseg_param_count", "vitseg_param_offset", "vitseg_cast_params_bf16",
    "vitseg_query_workspace", "vitseg_workspace_offset", "vitseg_forward", "vitseg_op_layernorm_f32",
    "vitseg_op_linear_f32", "vitseg_op_attention_f32", "vitseg_op_upsample_argmax", "vitseg_op_upsample_bwd",
    "vitseg_profile_enable", "vitseg_profile_collect", "vitseg_op_linear_bf16", "vitseg_op_attention_bf16",
    "vitseg_ce_scratch_bytes", "vitseg_ce_loss",
    "vitseg_train_workspace", "vitseg_forward_train", "vitseg_backward", "vitseg_adam_step",
    "vitseg_grad_bucket_count", "vitseg_grad_bucket_range",
    "vitseg_cast_params_f16", "vitseg_op_linear_f16", "vitseg_op_attention_f16", "vitseg_op_linear_f32x3", "vitseg_op_attention_f32x3", "vitseg_cast_params_split",
    "vitseg_resize_taps", "vitseg_resize_coeffs", "vitseg_nearest_index", "vitseg_preprocess_u8",
    "vitseg_resize_nearest_u8", "vitseg_eval_counts", "vitseg_paed_scratch_bytes", "vitseg_paed_multiclass_loss",
    "vitseg_op_gemm_f32", "vitseg_op_attention_bwd_f32", "vitseg_op_layernorm_bwd_f32", "vitseg_op_linear_h16_ex",
    "vitseg_op_wgrad_bf16", "vitseg_op_wgrad_bf16_scratch_floats", "vitseg_op_attention_bwd_bf16", "vitseg_attention_dropmask_bytes", "vitseg_attention_bwd_scratch_floats", "vitseg_op_colsum_scratch_floats",
    "vitseg_paed_binary_scratch_bytes", "vitseg_paed_binary_loss", "vitseg_op_linear_f32_ex", "vitseg_resize_nearest_i64",
    "vitseg_small_splits", "vitseg_op_linear_f32_small", "vitseg_op_linear_resln_f32_small", "vitseg_op_attention_f32_small", "vitseg_op_attention_bwd_f32_small", "vitseg_op_linear_h16_small", "vitseg_op_attention_h16_small", "vitseg_forward_route", "vitseg_op_layernorm_bwd_f32_small",
    "vitseg_dbg_linear_f32_small", "vitseg_adamw_step", "vitseg_op_dgrad_f32_small", "vitseg_op_wgrad_f32_small", "vitseg_op_layernorm_bwd_scratch_floats",
]
VERSION = 110   # include/vitseg.h VITSEG_VERSION this binding was written against
KERNEL_KINDS = ["gemm_bias", "gemm_gelu", "gemm_resadd", "gemm_patch", "gemm_conv3", "attention", "layernorm",
                "head1x1", "upsample", "train_gemm_fwd", "train_dgrad", "train_wgrad", "train_attn_fwd", "train_attn_bwd"]


class CConfig(C.Structure):
    _fields_ = [("num_classes", C.c_int32), ("patch_size", C.c_int32), ("hidden_size", C.c_int32),
                ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("image_size", C.c_int32),
                ("intermediate_size", C.c_int32), ("num_channels", C.c_int32), ("layer_norm_eps", C.c_float)]

    @classmethod
    def from_config(cls, cfg: ViTSegConfig) -> "CConfig":
        return cls(cfg.num_classes, cfg.patch_size, cfg.hidden_size, cfg.num_hidden_layers,
                   cfg.num_attention_heads, cfg.image_size, cfg.intermediate_size, cfg.num_channels,
                   cfg.layer_norm_eps)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m visiontransformer_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        l = C.CDLL(LIB_PATH)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        pcfg, psz = C.POINTER(CConfig), C.POINTER(C.c_size_t)
        l.vitseg_version.restype = i32
        l.vitseg_last_error.restype = C.c_char_p
        l.vitseg_set_option.argtypes = [C.c_char_p, C.c_longlong]
        l.vitseg_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_longlong)]
        l.vitseg_param_count.argtypes = [pcfg, psz]
        l.vitseg_param_offset.argtypes = [pcfg, i32, i32, psz, psz]
        l.vitseg_cast_params_bf16.argtypes = [vp, vp, sz, vp]
        l.vitseg_cast_params_f16.argtypes = [vp, vp, sz, vp]
        l.vitseg_cast_params_split.argtypes = [vp, vp, sz, vp]
        l.vitseg_query_workspace.argtypes = [pcfg, i32, i32, psz]
        l.vitseg_workspace_offset.argtypes = [pcfg, i32, i32, i32, psz, psz]
        l.vitseg_forward.argtypes = [pcfg, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]
        l.vitseg_op_layernorm_f32.argtypes = [vp, vp, vp, vp, i32, i32, C.c_float, vp]
        l.vitseg_op_linear_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_linear_f32_ex.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, C.c_float, C.c_uint32, C.c_uint32, vp]
        l.vitseg_op_attention_f32.argtypes = [vp, vp, i32, i32, i32, vp]
        l.vitseg_op_attention_f32_small.argtypes = [vp, vp, i32, i32, i32, vp]
        l.vitseg_small_splits.argtypes = [i32, i32]
        l.vitseg_op_layernorm_bwd_scratch_floats.argtypes = [i32, i32]
        l.vitseg_op_layernorm_bwd_scratch_floats.restype = sz
        l.vitseg_op_wgrad_f32_small.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        l.vitseg_op_dgrad_f32_small.argtypes = [vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]
        l.vitseg_dbg_linear_f32_small.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, vp]
        l.vitseg_op_linear_f32_small.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_linear_resln_f32_small.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, C.c_float, vp]
        l.vitseg_op_linear_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_attention_bf16.argtypes = [vp, vp, i32, i32, i32, vp]
        l.vitseg_op_linear_f16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_linear_f32x3.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_wgrad_bf16_scratch_floats.argtypes = [i32, i32, i32]
        l.vitseg_op_wgrad_bf16_scratch_floats.restype = sz
        l.vitseg_op_wgrad_bf16.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
        l.vitseg_op_linear_h16_ex.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, C.c_float,
                                              C.c_uint32, C.c_uint32, vp, vp, vp]
        l.vitseg_op_colsum_scratch_floats.argtypes = [i32, i32]
        l.vitseg_op_colsum_scratch_floats.restype = sz
        l.vitseg_op_attention_f16.argtypes = [vp, vp, i32, i32, i32, vp]
        l.vitseg_op_attention_f32x3.argtypes = [vp, vp, i32, i32, i32, vp]
        l.vitseg_op_upsample_argmax.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_upsample_bwd.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_ce_scratch_bytes.argtypes = [i32, i32]
        l.vitseg_ce_scratch_bytes.restype = sz
        l.vitseg_ce_loss.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, vp]
        f32 = C.c_float
        l.vitseg_train_workspace.argtypes = [pcfg, i32, i32, psz]
        l.vitseg_forward_train.argtypes = [pcfg, vp, vp, vp, i32, i32, f32, C.c_uint64, vp, vp, sz, vp]
        l.vitseg_backward.argtypes = [pcfg, vp, vp, vp, i32, i32, f32, C.c_uint64, vp, i32, vp, vp, vp, f32, vp, vp, sz, vp]
        l.vitseg_grad_bucket_count.argtypes = [pcfg]
        l.vitseg_resize_taps.argtypes = [i32, i32]
        l.vitseg_paed_scratch_bytes.argtypes = [i32, i32, i32, i32]
        l.vitseg_paed_scratch_bytes.restype = sz
        l.vitseg_paed_multiclass_loss.argtypes = [vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp]
        l.vitseg_paed_binary_scratch_bytes.argtypes = [i32, i32, i32]
        l.vitseg_paed_binary_scratch_bytes.restype = sz
        l.vitseg_paed_binary_loss.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        l.vitseg_resize_coeffs.argtypes = [i32, i32, vp, vp]
        l.vitseg_nearest_index.argtypes = [i32, i32, i32, vp]
        l.vitseg_preprocess_u8.argtypes = [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]
        l.vitseg_resize_nearest_u8.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, vp, i32, vp, vp]
        l.vitseg_resize_nearest_i64.argtypes = [vp, i32, i32, i32, vp, vp, i32, i32, i32, vp, vp]
        l.vitseg_eval_counts.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]
        l.vitseg_grad_bucket_range.argtypes = [pcfg, i32, psz, psz]
        l.vitseg_adam_step.argtypes = [vp, vp, vp, vp, sz, f32, f32, f32, f32, i32, f32, vp]
        l.vitseg_adamw_step.argtypes = [vp, vp, vp, vp, sz, f32, f32, f32, f32, f32, i32, f32, vp]
        l.vitseg_op_gemm_f32.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
        l.vitseg_op_attention_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
        l.vitseg_op_layernorm_bwd_f32_small.argtypes = [vp, vp, vp, sz, i32, vp, vp, vp, vp, vp, i32, i32, f32, vp, vp, f32, C.c_uint32, C.c_uint32, vp]
        l.vitseg_forward_route.argtypes = [pcfg, i32, i32]
        l.vitseg_op_attention_h16_small.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        l.vitseg_op_linear_h16_small.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]
        l.vitseg_op_attention_bwd_f32_small.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, f32, C.c_uint32, C.c_uint32, vp]
        l.vitseg_attention_bwd_scratch_floats.argtypes = [i32, i32, i32]
        l.vitseg_attention_bwd_scratch_floats.restype = sz
        l.vitseg_op_attention_bwd_bf16.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, C.c_float, C.c_uint32, C.c_uint32, vp, vp, vp]
        l.vitseg_attention_dropmask_bytes.argtypes = [i32, i32, i32]
        l.vitseg_attention_dropmask_bytes.restype = C.c_size_t
        l.vitseg_op_layernorm_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp]
        l.vitseg_profile_enable.argtypes = [i32]
        l.vitseg_profile_collect.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        for name in EXPORTS:
            getattr(l, name)  # raises AttributeError if the build is stale
        if l.vitseg_version() != VERSION:   # argument lists changed between versions: a stale .so would misread them
            raise RuntimeError(f"{LIB_PATH} is version {l.vitseg_version()}, this binding expects {VERSION}: rebuild it "
                               "(python -m visiontransformer_amd.build)")
        _lib = l
    return _lib


def check(rc: int) -> None:
    """Maps C status codes to the exception types the reference raises
    (ValueError for shapes the reference rejects, modeling_vit.py:63-68,152-156)."""
    if rc == OK:
        return
    msg = lib().vitseg_last_error().decode(errors="replace")
    if rc == ESHAPE:
        raise ValueError(msg)
    raise RuntimeError(f"libvitseg error {rc}: {msg}")


def set_option(name: str, value: int) -> None:
    """Dispatcher switch (include/vitseg.h vitseg_set_option): e.g. set_option("no_f32p", 1)."""
    check(lib().vitseg_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_longlong()
    check(lib().vitseg_get_option(name.encode(), C.byref(v)))
    return int(v.value)


class option:
    """`with _lib.option("no_p8", 1): ...` -- the switch is restored on exit."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def param_count(cfg: ViTSegConfig) -> int:
    n = C.c_size_t()
    check(lib().vitseg_param_count(C.byref(CConfig.from_config(cfg)), C.byref(n)))
    return n.value


def param_offset(cfg: ViTSegConfig, tensor: int, layer: int = 0):
    off, n = C.c_size_t(), C.c_size_t()
    check(lib().vitseg_param_offset(C.byref(CConfig.from_config(cfg)), tensor, layer, C.byref(off), C.byref(n)))
    return off.value, n.value


def query_workspace(cfg: ViTSegConfig, batch: int, precision: int) -> int:
    n = C.c_size_t()
    check(lib().vitseg_query_workspace(C.byref(CConfig.from_config(cfg)), batch, precision, C.byref(n)))
    return n.value


def forward_route(cfg: ViTSegConfig, batch: int, precision: int) -> str:
    """"small" (the small-batch route of csrc/small.hpp) or "large": which kernels vitseg_forward takes for this call."""
    rc = lib().vitseg_forward_route(C.byref(CConfig.from_config(cfg)), batch, precision)
    if rc < 0:
        check(rc)
    return "small" if rc == 1 else "large"


def workspace_offset(cfg: ViTSegConfig, batch: int, precision: int, buffer: int):
    off, n = C.c_size_t(), C.c_size_t()
    check(lib().vitseg_workspace_offset(C.byref(CConfig.from_config(cfg)), batch, precision, buffer,
                                        C.byref(off), C.byref(n)))
    return off.value, n.value


def train_workspace(cfg: ViTSegConfig, batch: int, precision: int) -> int:
    n = C.c_size_t()
    check(lib().vitseg_train_workspace(C.byref(CConfig.from_config(cfg)), batch, precision, C.byref(n)))
    return n.value


def grad_buckets(cfg: ViTSegConfig) -> list:
    """[(offset_floats, n_floats)] of the gradient arena in the order vitseg_backward finishes them
    (0 = final norm + seg_head, 1 .. L = layers L-1 .. 0, L + 1 = embeddings)."""
    c = CConfig.from_config(cfg)
    n = lib().vitseg_grad_bucket_count(C.byref(c))
    if n < 0:
        check(n)
    out = []
    for i in range(n):
        off, cnt = C.c_size_t(), C.c_size_t()
        check(lib().vitseg_grad_bucket_range(C.byref(c), i, C.byref(off), C.byref(cnt)))
        out.append((off.value, cnt.value))
    return out


def profile_enable(on: bool) -> None:
    check(lib().vitseg_profile_enable(1 if on else 0))


def profile_collect() -> dict:
    """kind -> dict(ms, launches, work) for every kernel kind recorded since profile_enable(True)."""
    out = {}
    for i, name in enumerate(KERNEL_KINDS):
        ms, n, w = C.c_double(), C.c_int64(), C.c_double()
        check(lib().vitseg_profile_collect(i, C.byref(ms), C.byref(n), C.byref(w)))
        out[name] = dict(ms=ms.value, launches=n.value, work=w.value)
    return out
