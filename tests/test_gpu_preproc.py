"""GPU (-m gpu): device pre-processing (f3) and evaluation counting (f4) through the C ABI, bit-exact against the oracle
and the Pillow / torch fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import make_golden_preproc as G
from oracle import preproc_oracle as O
from visiontransformer_amd.metrics import Evaluator
from visiontransformer_amd.preprocess import NEAREST_PIL, NEAREST_TORCH, Preprocessor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "preproc", "preproc.npz"))


@pytest.mark.parametrize("seed,H,W,S", G.RESIZE_CASES)
def test_resize_to_tensor_matches_pillow_fixture(seed, H, W, S):
    a = G.image(seed, H, W)
    x = Preprocessor(S, DEV).images(torch.from_numpy(a))
    assert x.shape == (1, 3, S, S) and x.dtype == torch.float32
    assert np.array_equal(x[0].cpu().numpy(), O.to_tensor(Z[f"resize.{seed}.out"]))   # bit-exact, /255 included


def test_resize_batch_ragged_and_large():
    rs = np.random.RandomState(3)
    pre = Preprocessor(224, DEV)
    batch = rs.randint(0, 256, size=(3, 150, 310, 3), dtype=np.uint8)       # a batch of equal-size frames
    x = pre.images(torch.from_numpy(batch)).cpu().numpy()
    for i in range(3):
        assert np.array_equal(x[i], O.preprocess_image(batch[i], 224))
    for H, W in [(224, 300), (300, 224), (224, 224), (1, 1), (2, 999), (1536, 2048)]:   # one axis equal, identity, tiny, photo
        a = rs.randint(0, 256, size=(H, W, 3), dtype=np.uint8)
        assert np.array_equal(pre.images(torch.from_numpy(a))[0].cpu().numpy(), O.preprocess_image(a, 224)), (H, W)
    with pytest.raises(ValueError):
        pre.images(torch.zeros(4, 4, 4, dtype=torch.uint8))


@pytest.mark.parametrize("seed,H,W,oh,ow,C", G.NEAREST_CASES)
def test_mask_resize_and_remap(seed, H, W, oh, ow, C):
    m = G.nearest_source(seed, H, W)
    pre = Preprocessor(224, DEV)
    got = pre.masks(torch.from_numpy(m), (oh, ow), NEAREST_PIL, dtype=torch.uint8)[0].cpu().numpy()
    assert np.array_equal(got, Z[f"nearest.{seed}.pil"])
    got = pre.masks(torch.from_numpy(m), (oh, ow), NEAREST_TORCH, dtype=torch.uint8)[0].cpu().numpy()
    assert np.array_equal(got, Z[f"nearest.{seed}.torch"])
    mapping = {int(v): int(v) * 7 % C for v in np.unique(m)}
    y = pre.masks(torch.from_numpy(m), (oh, ow), NEAREST_PIL, value_to_class=mapping)     # torch.long, as the dataset yields
    assert y.dtype == torch.long
    lut = np.zeros(256, np.uint8)
    for v, c in mapping.items():
        lut[v] = c
    assert np.array_equal(y[0].cpu().numpy(), lut[Z[f"nearest.{seed}.pil"]].astype(np.int64))


@pytest.mark.parametrize("seed,gs,ps,C", G.METRIC_CASES)
def test_eval_counts_and_metrics(seed, gs, ps, C):
    gt, pred = G.metric_pair(seed, gs, ps, C, lambda g: O.resize_nearest_lut(g, ps, ps, "pil"))
    ev = Evaluator(C, DEV)
    # a batch of two: the case and a shifted copy (different statistics per image)
    pred2 = np.roll(pred, 17, axis=1)
    counts = ev.counts(torch.from_numpy(np.stack([pred, pred2])), torch.from_numpy(np.stack([gt, gt]))).cpu().numpy()
    gt_r = O.resize_nearest_lut(gt, ps, ps, "pil")
    assert np.array_equal(counts[0], O.class_counts(pred, gt_r, 256))
    assert np.array_equal(counts[1], O.class_counts(pred2, gt_r, 256))
    m = ev.evaluate(torch.from_numpy(pred[None]), torch.from_numpy(gt[None]))[0]
    exp = Z[f"metric.{seed}.scalars"]
    assert np.array_equal(np.array([m["Accuracy"], m["Mean_IoU"], m["Mean_Dice"]]), exp, equal_nan=True)
    ref = O.image_metrics(pred, gt, C)
    for k in ("GT_Classes", "Pred_Classes", "Missing_Classes", "False_Positive_Classes"):
        assert m[k] == ref[k]


def test_eval_counts_full_size_properties():
    """BASELINE batch size (32 x 512 x 512): conservation laws of the counts -- every pixel is counted once per side,
    agreement never exceeds either side, identical inputs give 100 % -- and labels outside the class range (255 =
    ignore) are kept in the class sets exactly as np.unique would."""
    rs = np.random.RandomState(0)
    pred = torch.from_numpy(rs.randint(0, 17, size=(32, 512, 512), dtype=np.uint8))
    gt = torch.from_numpy(rs.randint(0, 17, size=(32, 256, 256), dtype=np.uint8))
    gt[:, :8] = 255
    ev = Evaluator(17, DEV)
    c = ev.counts(pred, gt).cpu().numpy()
    assert (c[:, 1].sum(1) == 512 * 512).all() and (c[:, 2].sum(1) == 512 * 512).all()
    assert (c[:, 0] <= c[:, 1]).all() and (c[:, 0] <= c[:, 2]).all()
    assert (c[:, 1, 255] > 0).all() and 255 in ev.evaluate(pred[:1], gt[:1])[0]["GT_Classes"]
    same = ev.evaluate(pred, pred)
    assert all(m["Accuracy"] == 100.0 and m["Mean_IoU"] == 1.0 and m["Mean_Dice"] == 1.0 for m in same)
    i = 5
    ref = O.image_metrics(pred[i].numpy(), gt[i].numpy(), 17)
    got = ev.evaluate(pred[i:i + 1], gt[i:i + 1])[0]
    assert got["Accuracy"] == ref["Accuracy"] and got["Mean_IoU"] == ref["Mean_IoU"] and got["Mean_Dice"] == ref["Mean_Dice"]


def test_predict_from_encoded_bytes_uses_device_preprocessing():
    """predict(): PNG bytes -> decode (host) -> Resize + ToTensor (device) -> model -> mask; equals the oracle's
    pre-processing followed by the same model, and the mask of a second call is identical (cached tables)."""
    import io
    from PIL import Image
    from visiontransformer_amd import synth
    from visiontransformer_amd.config import ViTSegConfig
    from visiontransformer_amd.model import ViTSegmentationModel
    from visiontransformer_amd.predict import predict
    cfg = ViTSegConfig(3, 16, 192, 2, 3, image_size=224)
    m = ViTSegmentationModel(3, 16, 192, 2, 3, image_size=224, device=DEV).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=5).items()})
    a = np.random.RandomState(11).randint(0, 256, size=(333, 500, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a, "RGB").save(buf, format="PNG")
    colors = np.array([[0, 0, 0], [255, 0, 0], [0, 255, 0]], np.uint8)
    mask, rgb, logits = predict(buf.getvalue(), m, index_to_color=colors, return_logits=True)
    x = torch.from_numpy(O.preprocess_image(a, 224))[None].to(DEV)
    with torch.no_grad():
        ref_mask, ref_logits = m.predict_mask(x, return_logits=True)
    assert np.array_equal(mask, ref_mask[0].cpu().numpy()) and np.array_equal(logits, ref_logits[0].cpu().numpy())
    assert rgb.shape == (224, 224, 3) and np.array_equal(rgb, colors[mask])
    assert np.array_equal(predict(a, m), mask)


def test_worker_end_to_end_on_the_gpu():
    """Row f2 with the real network behind it: images of different sizes are queued over HTTP, batched into one forward
    on the MI355X and come back as colourised PNGs equal to predict() on each image alone."""
    import io
    import threading
    import time
    from PIL import Image
    from test_worker_cpu import TOKEN, FakeBackend, _png, _post
    from visiontransformer_amd import synth
    from visiontransformer_amd.config import ViTSegConfig
    from visiontransformer_amd.predict import predict
    from visiontransformer_amd.worker import Worker, gpu_slot
    slot = gpu_slot((16, 192, 2, 3), 3, image_size=224, device=DEV)
    cfg = ViTSegConfig(3, 16, 192, 2, 3, image_size=224)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=5).items()}
    from visiontransformer_amd.model import ViTSegmentationModel
    ref_model = ViTSegmentationModel(3, 16, 192, 2, 3, image_size=224, device=DEV).eval()
    ref_model.load_state_dict(sd)
    slot.model.model.load_state_dict(sd)   # same weights as the stand-alone model (gpu_slot starts from random init)
    be = FakeBackend()
    w = Worker({1: slot}, be.url, TOKEN, max_batch=8, batch_wait_s=0.05)
    srv = w.serve("127.0.0.1", 0)
    threading.Thread(target=srv.serve_forever, daemon=True).start()
    url = f"http://127.0.0.1:{srv.server_address[1]}"
    rs = np.random.RandomState(4)
    imgs = [rs.randint(0, 256, size=(h, wd, 3), dtype=np.uint8) for h, wd in [(224, 224), (100, 333), (480, 640), (50, 50), (224, 300)]]
    try:
        for i, a in enumerate(imgs):
            assert _post(url + "/enqueue/", {"job_id": f"j{i}", "vision_model_id": "1"},
                         {"input_image": ("x.png", _png(a), "image/png")})[0] == 202
        for _ in range(500):
            if len(be.done) == len(imgs):
                break
            time.sleep(0.02)
        assert len(be.done) == len(imgs) and w.stats["failed"] == 0
        for i, a in enumerate(imgs):
            got = np.array(Image.open(io.BytesIO(be.done[f"j{i}"][1])).convert("RGB"))
            assert np.array_equal(got, slot.palette[predict(a, ref_model)])
    finally:
        srv.shutdown()
        w.stop()
        be.srv.shutdown()


# ---------------------------------------------------------------- row f1: soft PAED loss for C classes, fused
def test_paed_multiclass_fused_matches_reference_golden():
    """csrc/paed_loss.hip against values and gradients produced by the REAL paed_loss_multiclass_soft
    (tests/golden/paed/paed_losses.npz, oracle/make_golden_paed.py)."""
    from oracle.make_golden_paed import paed_inputs
    from visiontransformer_amd import paed
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "paed", "paed_losses.npz"))
    logits, y, *_ = paed_inputs()
    lg = logits.to(DEV).requires_grad_(True)
    loss = paed.paed_multiclass_loss_fused(lg, y.to(DEV))
    loss.backward()
    assert abs(loss.item() - float(G["multiclass.loss"][0])) < 2e-7 * max(1.0, abs(float(G["multiclass.loss"][0])))
    ref = G["multiclass.grad"]
    assert np.abs(lg.grad.cpu().numpy() - ref).max() < 2e-6 * np.abs(ref).max() + 1e-12


@pytest.mark.parametrize("B,C,H,W,sigma,pen,u8", [(2, 17, 224, 224, 3, True, False), (1, 5, 50, 70, 3, True, True),
                                                  (3, 4, 33, 21, 2, False, False), (1, 2, 8, 8, 3, True, True)])
def test_paed_multiclass_fused_matches_torch_autograd(B, C, H, W, sigma, pen, u8):
    """Shapes the golden file does not hold (non-square, maps smaller than the 19-tap window, class_penalty off,
    uint8 targets) against fp64 autograd through the oracle's restatement of the reference function
    (oracle/paed_oracle.py, pinned to the reference's own outputs by tests/test_paed_cpu.py)."""
    import torch.nn.functional as F
    from oracle import paed_oracle as PO
    from visiontransformer_amd import paed
    g = torch.Generator().manual_seed(B * 1000 + C)
    logits = torch.randn(B, C, H, W, generator=g) * 2
    y = torch.randint(0, C, (B, H, W), generator=g)
    ld = logits.double().requires_grad_(True)
    ref = PO.multiclass_soft_paed(F.one_hot(y, C).permute(0, 3, 1, 2).double(), torch.softmax(ld, dim=1), sigma=sigma,
                                  class_penalty=pen)
    ref.backward()
    lg = logits.to(DEV).requires_grad_(True)
    yt = (y.to(torch.uint8) if u8 else y).to(DEV)
    loss = paed.paed_multiclass_loss_fused(lg, yt, sigma=sigma, class_penalty=pen)
    (loss * 3.0).backward()                      # the upstream gradient scales it
    assert abs(loss.item() - ref.item()) < 3e-6 * max(1.0, abs(ref.item()))
    gr = ld.grad.numpy() * 3.0
    assert np.abs(lg.grad.cpu().numpy() - gr).max() < 1e-5 * np.abs(gr).max() + 1e-12
    with torch.no_grad():                        # no gradient requested: loss only
        assert abs(paed.paed_multiclass_loss_fused(logits.to(DEV), yt, sigma=sigma, class_penalty=pen).item() - ref.item()) \
            < 3e-6 * max(1.0, abs(ref.item()))


# ---------------------------------------------------------------- row f1, second half: the binary PAED trainer's loss tail
def test_paed_binary_fused_matches_reference_golden():
    """csrc/paed_binary.hip (sigmoid + BCE + 0.1 Dice + 5 |soft PAED|, value and gradient) against the outputs of the REAL
    reference methods dice_loss / paed_loss_soft + F.binary_cross_entropy (model/PAED/classes.py:608-681; fixture from
    oracle/make_golden_paed.py)."""
    from oracle.make_golden_paed import paed_inputs
    from visiontransformer_amd import paed
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "paed", "paed_losses.npz"))
    _, _, blogits, bmask, sdf_ext, sdf_int = paed_inputs()
    lg = blogits.to(DEV).requires_grad_(True)
    loss, terms = paed.paed_binary_loss_fused(lg, bmask.to(DEV), sdf_ext[:, 0].to(DEV), sdf_int[:, 0].to(DEV))
    loss.backward()
    t = terms.cpu().numpy()
    assert abs(t[3] - float(G["binary.paed"][0])) < 2e-6 and abs(t[2] - float(G["binary.dice"][0])) < 2e-6
    assert abs(loss.item() - float(G["binary.total"][0])) < 2e-6 * max(1.0, abs(float(G["binary.total"][0])))
    ref = G["binary.grad"]
    assert np.abs(lg.grad.cpu().numpy() - ref).max() < 5e-6 * np.abs(ref).max() + 1e-12
    p = torch.sigmoid(blogits)
    b = (p > 0.5).float()
    assert t[4] == float((b * bmask).sum()) and t[5] == float((b * (1 - bmask)).sum())
    assert t[6] == float(((1 - b) * bmask).sum()) and t[7] == float((b == bmask).sum())


@pytest.mark.parametrize("B,H,W,hs,ws", [(2, 224, 224, 64, 64), (1, 50, 70, 20, 31), (3, 33, 21, 33, 21), (1, 96, 96, 128, 128)])
def test_paed_binary_fused_matches_torch_autograd(B, H, W, hs, ws):
    """Shapes the fixture does not hold (non-square, tiles with ragged edges, SDFs larger than the prediction) against fp64
    autograd through the oracle's restatement of the reference methods (oracle/paed_oracle.py, pinned to the reference's own
    outputs by tests/test_paed_cpu.py)."""
    from oracle import paed_oracle as PO
    from visiontransformer_amd import paed
    g = torch.Generator().manual_seed(B * 100 + H)
    z = torch.randn(B, 1, H, W, generator=g) * 2
    m = (torch.rand(B, 1, H, W, generator=g) > 0.55).float()
    se, si = torch.rand(B, hs, ws, generator=g) * 5, torch.rand(B, hs, ws, generator=g) * 3
    zd = z.double().requires_grad_(True)
    pa = PO.binary_soft_paed(se.unsqueeze(1).double(), si.unsqueeze(1).double(), torch.sigmoid(zd))
    ref = PO.binary_total(zd, m.double(), se.unsqueeze(1).double(), si.unsqueeze(1).double())
    ref.backward()
    lg = z.to(DEV).requires_grad_(True)
    loss, terms = paed.paed_binary_loss_fused(lg, m.to(DEV), se.to(DEV), si.to(DEV))
    (loss * 2.0).backward()
    assert abs(loss.item() - ref.item()) < 5e-6 * max(1.0, abs(ref.item()))
    assert abs(float(terms[3]) - pa.item()) < 5e-6
    gr = zd.grad.numpy() * 2.0
    assert np.abs(lg.grad.cpu().numpy() - gr).max() < 2e-5 * np.abs(gr).max() + 1e-12


# ---------------------------------------------------------------- the reference's script entry points (model/CE, model/PAED)
def test_reference_entry_points_run_end_to_end():
    """tools/entrypoints_smoke.sh: testViTModel / trainCurrentViTmodel / createViTmodel / datasetTestViTmodel (CE) and
    ViTscript / ViTscriptUp (resume from the checkpoint it just wrote) / ViTscriptTest (PAED) on synthetic data, each in
    its own process from a scratch directory; the evaluation drivers must leave the CSV schema compareModels.py reads."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "entrypoints_smoke.sh")], cwd=root, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "entry points ok" in r.stdout
    assert "Model_ID,Model_Name,Patch_Size,Hidden_Size,Layers,Heads,Batch_Num,Image_Idx,Accuracy,Mean_IoU,Mean_Dice" in r.stdout


# ---------------------------------------------------------------- a13: target resize on the device path
@pytest.mark.parametrize("H,W,oh,ow", [(256, 256, 224, 224), (256, 256, 512, 512), (97, 301, 224, 224), (224, 224, 224, 224)])
def test_resize_target_gather_kernel_is_f_interpolate_nearest(H, W, oh, ow):
    """LightningViTModel._resize_target on GPU tensors (one gather kernel, vitseg_resize_nearest_i64 / _u8 with the
    mode-1 tables) == F.interpolate(y[:, None].float(), size, mode='nearest').long() (model/CE/classes.py:273-274), bit for
    bit, for int64 and uint8 class maps and both result types."""
    import torch.nn.functional as F
    from visiontransformer_amd.lightning import LightningViTModel
    g = torch.Generator().manual_seed(H * 7 + ow)
    y = torch.randint(0, 17, (3, H, W), generator=g)
    ref = F.interpolate(y.unsqueeze(1).float(), size=(oh, ow), mode="nearest").squeeze(1).long()
    from oracle import vitseg_oracle as VO
    assert torch.equal(ref, VO.resize_target(y, (oh, ow)))                      # the oracle's restatement agrees too
    lm = LightningViTModel(17, 16, 192, 1, 3, image_size=224, device=DEV)
    out = lm._resize_target(y.to(DEV), (oh, ow))
    assert out.dtype == torch.long and torch.equal(out.cpu(), ref)
    out8 = lm._resize_target(y.to(torch.uint8).to(DEV), (oh, ow), dtype=torch.uint8)
    assert out8.dtype == torch.uint8 and torch.equal(out8.cpu().long(), ref)
    assert torch.equal(lm._resize_target(y.to(DEV), (oh, ow), dtype=torch.uint8).cpu().long(), ref)
    assert torch.equal(lm._resize_target(y, (oh, ow)), ref)                     # CPU tensors: host-side path as the reference
