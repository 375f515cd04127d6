"""CPU: the oracle restatement (oracle/vitseg_oracle.py) against the golden vectors
captured from the REAL reference class (oracle/make_golden.py).  This is the pin
that lets the GPU parity tests trust the oracle on the GPU box, where neither
/root/reference nor `transformers` is consulted."""
import numpy as np
import pytest
import torch

from oracle import vitseg_oracle as O
from util import CASES, Golden

# fp32 restatement vs fp32 reference: same maths, different op order (matmul vs sdpa/mkldnn conv)
TOL_STAGE = 2e-5
TOL_LOGITS = 2e-5


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference(name):
    g = Golden(name)
    torch.set_num_threads(8)
    stages = {}
    with torch.no_grad():
        logits = O.forward(g.images(), g.state_dict(), g.cfg, stages)
    for st in ["embeddings", "ln1_0", "q_0", "k_0", "v_0", "ctx_0", "attn_res_0", "mlp_0", "layer_0",
               "last_hidden_state", "lowres_logits"]:
        err, scale = g.max_abs_err("stage." + st, stages[st])
        assert err <= TOL_STAGE * max(1.0, scale), (st, err, scale)
    full = g.z["lowres_logits.full"]
    scale = max(1.0, float(np.abs(full).max()))
    assert np.abs(stages["lowres_logits"].numpy() - full).max() <= TOL_LOGITS * scale
    err, sc = g.max_abs_err("logits", logits)
    assert err <= TOL_LOGITS * max(1.0, sc), err
    assert g.checksum_rel_err("logits", logits) < 1e-5
    # masks: sigmoid -> first-max argmax; must agree wherever the reference decision is not fragile
    mask = O.predict_mask(logits).numpy()
    ref = g.mask()
    bad = (mask != ref) & ~g.fragile()
    assert bad.sum() == 0, int(bad.sum())


@pytest.mark.parametrize("name", CASES)
def test_decoder_tail_bit_exact_given_lowres(name):
    """Given identical low-res logits, upsample + sigmoid + argmax must reproduce the
    reference mask on every pixel whose sigmoid values do not tie (exact fp32 restatement
    of ATen's bilinear arithmetic order)."""
    g = Golden(name)
    z = torch.from_numpy(g.z["lowres_logits.full"])
    S = g.cfg.image_size
    up = O.upsample_bilinear(z, (S, S))
    ref_up = torch.nn.functional.interpolate(z, size=(S, S), mode="bilinear", align_corners=False)
    assert torch.equal(up, ref_up), float((up - ref_up).abs().max())
    assert np.array_equal(O.predict_mask(up).numpy(), g.mask())


def test_sigmoid_restatement_is_atens():
    """O.sigmoid_aten (Sleef u10 exp restated + exact add / divide) against torch.sigmoid, bit for bit, on sizes that
    are whole vectors per thread chunk (the scalar tail of a chunk goes through glibc's expf instead)."""
    torch.manual_seed(0)
    for scale in (0.5, 2.0, 8.0, 20.0, 60.0, 200.0):
        x = (torch.randn(1 << 20) * scale).float()
        assert torch.equal(O.sigmoid_aten(x).view(torch.int32), torch.sigmoid(x).view(torch.int32)), scale
    edge = torch.tensor([0.0, -0.0, 88.0, -88.0, 103.9, -103.9, 17.32868, 17.4, 1e-8, -1e-8] * 32, dtype=torch.float32)
    assert torch.equal(O.sigmoid_aten(edge).view(torch.int32), torch.sigmoid(edge).view(torch.int32))


def test_mask_stable_flags_only_flippable_pixels():
    z = torch.tensor([[[[0.0]], [[1e-3]], [[18.0]], [[19.0]]]])           # classes 2 and 3 saturate to 1.0f: first wins
    assert int(O.predict_mask(z)) == 2 and bool(O.mask_stable(z, 1e-5))
    z = torch.tensor([[[[1.0]], [[1.0 + 2e-6]]]])                           # class 1 wins by 4e-7: an error of 1e-5 flips it
    assert int(O.predict_mask(z)) == 1 and not bool(O.mask_stable(z, 1e-5))
    z = torch.tensor([[[[10.0]], [[10.0 + 2e-6]]]])                         # an fp32 sigmoid tie that stays a tie: first wins
    assert int(O.predict_mask(z)) == 0 and bool(O.mask_stable(z, 1e-5))


def test_resize_target_matches_aten():
    y = torch.randint(0, 5, (3, 256, 256))
    for S in (224, 512, 100, 256):
        ref = torch.nn.functional.interpolate(y[:, None].float(), size=(S, S), mode="nearest").squeeze(1).long()
        assert torch.equal(O.resize_target(y, (S, S)), ref)


@pytest.mark.parametrize("name", [c for c in CASES if Golden(c).has("grad.seg_head.2.weight")])
def test_training_step_matches_reference(name):
    g = Golden(name)
    torch.set_num_threads(8)
    sd = g.state_dict()
    y = g.targets()
    assert np.array_equal(O.resize_target(y, (g.cfg.image_size,) * 2).numpy().astype(np.uint8),
                          g.z["train.target_resized"])
    loss, grads = O.training_step(g.images(), y, sd, g.cfg)
    assert abs(loss.item() - g.z["train.loss"][0]) < 2e-6
    for key in [k[5:-4] for k in g.z.files if k.startswith("grad.") and k.endswith(".idx")]:
        err, scale = g.max_abs_err("grad." + key, grads[key])
        assert err <= 2e-4 * scale + 1e-9, (key, err, scale)
    # one Adam(lr=1e-5) step from zero state: delta = -lr * g / (|g| + eps*sqrt(1-b2)) ...
    for key in [k[6:-4] for k in g.z.files if k.startswith("adam1.") and k.endswith(".idx")]:
        p = sd[key]
        gr = grads[key]
        newp, _, _ = O.adam_step(p, gr, torch.zeros_like(p), torch.zeros_like(p), step=1)
        err, scale = g.max_abs_err("adam1." + key, newp - p)
        # a sign flip of a ~0 gradient moves the update by 2*lr; allow a handful of such entries
        assert err <= 2.1e-5, (key, err)


def test_reference_error_behaviour():
    g = Golden("tiny16_224_c2")
    sd = g.state_dict()
    with pytest.raises(ValueError):
        O.forward(torch.zeros(1, 3, 256, 256), sd, g.cfg)
    with pytest.raises(ValueError):
        O.forward(torch.zeros(1, 1, 224, 224), sd, g.cfg)
