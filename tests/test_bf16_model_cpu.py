"""CPU: the fp64 rounding model of the bf16 training step (oracle/bf16_model.py) is the oracle's training step once its
roundings are switched off -- i.e. its hand-written backward formulas (attention core with dropout, saved GELU
derivative, patch embedding) are the derivatives autograd finds for the restated reference -- and a bf16-sized
perturbation of it once they are on."""
import torch

from dropout_ref import Masks
from oracle import bf16_model as BM
from oracle import vitseg_oracle as O
from visiontransformer_amd import synth
from visiontransformer_amd.config import ViTSegConfig


def _case(L=2):
    cfg = ViTSegConfig(3, 16, 128, L, 2, image_size=96)
    sd = {k: torch.from_numpy(v).double() for k, v in synth.make_state_dict(cfg, seed=5, perturb=True).items()}
    x = torch.from_numpy(synth.make_images(cfg, 2, seed=4)).double()
    y = torch.from_numpy(synth.make_targets(cfg, 2, seed=4, size=96))
    return cfg, sd, x, y


def _exact(cfg, sd, x, y, drop):
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss = O.ce_loss(O.forward(x, leaf, cfg, drop=drop), y)
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaf.items()}


def test_rounding_model_without_roundings_is_the_oracle(monkeypatch):
    cfg, sd, x, y = _case()
    for drop in (None, Masks(0.1, 0x1234567, 2, cfg.num_patches, cfg.num_attention_heads)):
        loss_ref, g_ref = _exact(cfg, sd, x, y, drop)
        monkeypatch.setattr(BM, "rb", lambda t: t)
        loss, g = BM.training_step(x, y, sd, cfg, drop)
        monkeypatch.undo()
        assert abs(float(loss) - float(loss_ref)) < 1e-12
        for k, r in g_ref.items():
            assert (g[k] - r).abs().max().item() <= 1e-10 * max(1.0, r.abs().max().item()), k


def test_rounding_model_is_a_bf16_sized_perturbation():
    cfg, sd, x, y = _case()
    loss_ref, g_ref = _exact(cfg, sd, x, y, None)
    loss, g = BM.training_step(x, y, sd, cfg)
    assert 0 < abs(float(loss) - float(loss_ref)) < 5e-3
    rels = {k: float((g[k] - r).norm() / r.norm()) for k, r in g_ref.items() if float(r.norm()) > 1e-9}
    assert max(rels.values()) < 0.1 and min(rels.values()) > 1e-5, rels
    # round to nearest, ties to even (8 significand bits: ulp(1) = 2^-7)
    assert BM.rb(torch.tensor([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8, 1.0 + 3 * 2.0 ** -9], dtype=torch.float64)).tolist() == \
        [1.0, 1.015625, 1.0078125]
