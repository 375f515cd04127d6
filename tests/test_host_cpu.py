"""CPU (-m "not gpu"): host logic + the C-ABI library loads and exports every symbol of include/vitseg.h.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from visiontransformer_amd import _lib, params, synth
from visiontransformer_amd.config import ViTSegConfig, vit_base16, vit_tiny16
from visiontransformer_amd.model import ViTSegmentationModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "vitseg.h")).read()
    declared = set(re.findall(r"\b(vitseg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    l = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(l, name), name
    assert _lib.lib().vitseg_version() == _lib.VERSION == 110


def test_param_arena_layout_is_disjoint_and_complete():
    cfg = vit_tiny16()
    total = _lib.param_count(cfg)
    spans = []
    for t in range(_lib.T_COUNT):
        layers = range(cfg.num_hidden_layers) if _lib.T_LN1_W <= t <= _lib.T_B2 else [0]
        for l in layers:
            off, n = _lib.param_offset(cfg, t, l)
            assert off % 64 == 0 and n > 0
            spans.append((off, off + n))
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= total
    n_ref = sum(int(np.prod(s)) for s in synth.param_shapes(cfg).values())
    assert sum(b - a for a, b in spans) == n_ref == 16_649_090 - 192 * 192 - 192  # minus the unused pooler


def test_config_errors_map_to_value_error():
    with pytest.raises(ValueError):
        _lib.param_count(ViTSegConfig(2, 16, 100, 2, 2))  # head_dim 50: unsupported by this build
    with pytest.raises(ValueError):
        ViTSegConfig(2, 16, 768, 12, 7)  # hidden not a multiple of heads (HF raises the same)


def test_state_dict_roundtrip_and_key_schemas():
    cfg = ViTSegConfig(3, 16, 192, 2, 3)
    m = ViTSegmentationModel(3, 16, 192, 2, 3)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=9).items()}
    m.load_state_dict(sd)
    out = m.state_dict()
    assert list(out.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(out[k], sd[k]), k
    # kernel-native layouts inside the arena
    off, n = _lib.param_offset(cfg, _lib.T_WQKV, 1)
    wqkv = m.arena.data[off:off + n].view(3 * 192, 192)
    assert torch.equal(wqkv[192:384], sd["backbone.layers.1.attention.k_proj.weight"])
    off, n = _lib.param_offset(cfg, _lib.T_HEAD0_W, 0)
    w0 = m.arena.data[off:off + n].view(256, 3, 3, 192)
    assert torch.equal(w0, sd["seg_head.0.weight"].permute(0, 2, 3, 1))
    # legacy transformers-4 names + Lightning prefix + pooler keys
    legacy = {}
    for k, v in sd.items():
        k2 = re.sub(r"backbone\.layers\.(\d+)\.attention\.q_proj", r"backbone.encoder.layer.\1.attention.attention.query", k)
        k2 = re.sub(r"backbone\.layers\.(\d+)\.attention\.k_proj", r"backbone.encoder.layer.\1.attention.attention.key", k2)
        k2 = re.sub(r"backbone\.layers\.(\d+)\.attention\.v_proj", r"backbone.encoder.layer.\1.attention.attention.value", k2)
        k2 = re.sub(r"backbone\.layers\.(\d+)\.attention\.o_proj", r"backbone.encoder.layer.\1.attention.output.dense", k2)
        k2 = re.sub(r"backbone\.layers\.(\d+)\.mlp\.fc1", r"backbone.encoder.layer.\1.intermediate.dense", k2)
        k2 = re.sub(r"backbone\.layers\.(\d+)\.mlp\.fc2", r"backbone.encoder.layer.\1.output.dense", k2)
        k2 = re.sub(r"backbone\.layers\.(\d+)\.layernorm", r"backbone.encoder.layer.\1.layernorm", k2)
        legacy["model." + k2] = v
    legacy["model.backbone.pooler.dense.weight"] = torch.zeros(192, 192)
    legacy["model.backbone.pooler.dense.bias"] = torch.zeros(192)
    m2 = ViTSegmentationModel(3, 16, 192, 2, 3)
    m2.load_state_dict(legacy)
    assert torch.equal(m2.arena.data, m.arena.data)
    with pytest.raises(RuntimeError):
        m2.load_state_dict({"bogus.weight": torch.zeros(1)})


def test_forward_rejects_what_the_reference_rejects_and_never_falls_back_to_cpu():
    m = ViTSegmentationModel(2, 16, 192, 1, 3).eval()
    with pytest.raises(ValueError, match="doesn't match model"):
        m(torch.zeros(1, 3, 256, 256))
    with pytest.raises(ValueError, match="channel dimension"):
        m(torch.zeros(1, 1, 224, 224))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 224, 224))


def test_synth_is_deterministic():
    cfg = vit_tiny16()
    a = synth.make_images(cfg, 2, seed=0)
    b = synth.make_images(cfg, 1, seed=0, first_image=1)
    assert np.array_equal(a[1], b[0]) and a.min() >= 0 and a.max() < 1
    u = synth.uniform01(3, "x", 4)
    assert np.allclose(u, synth.uniform01(3, "x", 4)) and not np.allclose(u, synth.uniform01(4, "x", 4))
    w = synth.make_state_dict(cfg, seed=1)["backbone.layers.0.mlp.fc1.weight"]
    assert abs(w.std() - 0.02) < 5e-4 and abs(w.mean()) < 1e-4
    assert vit_base16().forward_flops_per_image() / 1e9 == pytest.approx(217.7, abs=0.2)


def test_dropout_generator_statistics():
    """The counter-based dropout masks (tests/dropout_ref.py restates csrc/common.hpp): keep rate 1 - p and no visible
    correlation between neighbouring keys, neighbouring queries or neighbouring layers."""
    from dropout_ref import Masks
    m = Masks(0.1, 0x5EED1234ABCD, B=2, Np=255, A=3)
    a = (m.attn(0, (2, 3, 256, 256)).numpy() > 0).astype(np.float64)          # keep indicators
    assert abs(a.mean() - 0.9) < 2e-3
    def corr(x, y):
        x, y = x.ravel() - x.mean(), y.ravel() - y.mean()
        return float((x * y).mean() / (x.std() * y.std()))
    assert abs(corr(a[..., :-1], a[..., 1:])) < 0.01        # neighbouring keys (the two halves of one hash)
    assert abs(corr(a[..., :-2], a[..., 2:])) < 0.01        # next pair
    assert abs(corr(a[:, :, :-1], a[:, :, 1:])) < 0.01      # neighbouring queries
    for lag in (3, 4, 8, 32, 33):                           # further keys / queries, the two diagonals
        assert abs(corr(a[..., :-lag], a[..., lag:])) < 0.01 and abs(corr(a[:, :, :-lag], a[:, :, lag:])) < 0.01
    assert abs(corr(a[:, :, :-1, :-1], a[:, :, 1:, 1:])) < 0.01 and abs(corr(a[:, :, :-1, 1:], a[:, :, 1:, :-1])) < 0.01
    assert abs(a.mean(axis=(0, 1, 2)).std() - np.sqrt(0.09 / (2 * 3 * 256))) < 3e-3   # per-key keep rates scatter like iid bits
    b = (m.attn(1, (2, 3, 256, 256)).numpy() > 0).astype(np.float64)
    assert abs(corr(a, b)) < 0.01                           # next layer
    r = (m.rows(0, 3, (2, 256, 192)).numpy() > 0).astype(np.float64)
    assert abs(r.mean() - 0.9) < 3e-3 and abs(corr(r[..., :-1], r[..., 1:])) < 0.01
    assert abs(r.mean(axis=(0, 1)).std()) < 0.02            # no column that is dropped (or kept) systematically


def test_dispatcher_options_table():
    """vitseg_set_option / vitseg_get_option (include/vitseg.h): names are case-insensitive, unknown names are an error with a
    message, `_lib.option` restores the previous value, and the launch path's switches start from the environment once."""
    import subprocess
    import sys
    for name in ("no_f32p", "no_p8", "no_h16p", "no_ragged_p8", "no_dropmask", "upsample_global", "f32p_noinl", "gn", "no_mask2"):
        assert _lib.get_option(name) == 0, name
    assert _lib.get_option("dropw_limit_mb") == -1 and _lib.get_option("bf16_tiles") == 0
    with _lib.option("NO_P8", 1):
        assert _lib.get_option("no_p8") == 1
        with _lib.option("gn", 8):
            assert _lib.get_option("GN") == 8
        assert _lib.get_option("gn") == 0
    assert _lib.get_option("no_p8") == 0
    with pytest.raises(RuntimeError, match="unknown option"):
        _lib.set_option("no_such_switch", 1)
    # initial values come from VITSEG_<NAME> as the library is loaded (a fresh process)
    code = ("from visiontransformer_amd import _lib; "
            "print(_lib.get_option('no_h16p'), _lib.get_option('bf16_tiles'), _lib.get_option('dropw_limit_mb'))")
    env = dict(os.environ, VITSEG_NO_H16P="1", VITSEG_BF16_TILES="large", VITSEG_DROPW_LIMIT_MB="64")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "2", "64"], out
    # a numeric value is taken as is (=0 leaves a switch OFF, as vitseg_set_option(name, 0) does); empty / non-numeric = 1
    code = ("from visiontransformer_amd import _lib; "
            "print(*[_lib.get_option(n) for n in ('no_p8', 'no_mask2', 'no_small', 'no_f32p', 'small_max_rows', 'small_variant')])")
    env = dict(os.environ, VITSEG_NO_P8="0", VITSEG_NO_MASK2="", VITSEG_NO_SMALL="yes", VITSEG_NO_F32P="1", VITSEG_SMALL_MAX_ROWS="4096")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "1", "1", "1", "4096", "0"], out


def test_small_route_summation_rules_are_functions_of_the_shape():
    """csrc/small.hpp: the chunk count of a reduction (vitseg_small_splits) -- host arithmetic, no GPU needed: wide outputs are one
    chunk; o_proj / fc2 / the patch embedding / the activation gradients of the reference's three widths get the documented counts."""
    f = _lib.lib().vitseg_small_splits
    assert [f(2304, 768), f(3072, 768), f(1536, 512), f(3072, 1024)] == [1, 1, 1, 1]           # QKV, fc1: N > K
    assert [f(768, 768), f(768, 3072), f(512, 512), f(512, 3072), f(1024, 1024), f(1024, 3072)] == [3, 6, 2, 6, 4, 6]
    assert [f(768, 2304), f(512, 1536), f(1024, 3072)] == [6, 3, 6]                               # dH1 = dQKV . Wqkv
    assert [f(192, 192), f(192, 3072), f(192, 576), f(768, 100)] == [1, 6, 3, 1]                   # Tiny/16; K % 32 != 0: one chunk


def test_forward_route_is_host_arithmetic_on_the_shape():
    """vitseg_forward_route (include/vitseg.h): which kernels a forward takes, decided from (configuration, batch, precision)
    alone -- the reference's regime on the small-batch route, the headline on the large-batch kernels; no GPU needed."""
    from visiontransformer_amd.config import vit_base16
    b224 = ViTSegConfig(17, 16, 768, 12, 12, image_size=224)
    r = _lib.forward_route
    assert [r(b224, b, _lib.F32) for b in (1, 4, 64, 83, 84)] == ["small"] * 4 + ["large"]          # 197 tokens: < 16 384 rows
    assert r(vit_base16(), 15, _lib.F32) == "small" and r(vit_base16(), 16, _lib.F32) == "large"      # 1025 tokens (512x512)
    assert r(vit_base16(), 32, _lib.F32) == "large" and r(b224, 4, _lib.F32X3) == "large"
    assert [r(b224, b, _lib.BF16) for b in (1, 4, 16, 17)] == ["small", "small", "small", "large"]     # 16-bit form: < 3 200 rows
    assert r(vit_base16(), 1, _lib.BF16) == "large"                                                     # 1025 tokens: no 16-bit form
    p8 = ViTSegConfig(17, 8, 768, 12, 12, image_size=224)
    assert [r(p8, b, _lib.F16) for b in (1, 3, 4)] == ["small", "small", "large"]                      # 785 tokens: < 2 400 rows
    with _lib.option("no_small", 1):
        assert r(b224, 4, _lib.F32) == "large"
    with pytest.raises(ValueError):
        r(ViTSegConfig(2, 16, 100, 2, 2), 1, _lib.F32)
