"""CPU: the pre-/post-processing oracle (oracle/preproc_oracle.py) against fixtures produced by the real Pillow / torch
and by the reference's own metric statements (oracle/make_golden_preproc.py), plus the host-side logic of rows f3/f4:
the library's coefficient / index tables and the metric formulas on integer counts."""
import os
import zlib

import numpy as np
import pytest

from oracle import make_golden_preproc as G
from oracle import preproc_oracle as O

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "preproc", "preproc.npz"))


def _crc(a):
    return zlib.crc32(a.tobytes())


@pytest.mark.parametrize("seed,H,W,S", G.RESIZE_CASES)
def test_oracle_bilinear_resize_matches_pillow_fixture(seed, H, W, S):
    a = G.image(seed, H, W)
    assert _crc(a) == int(Z[f"resize.{seed}.crc"][0]), "input generator drifted"
    assert np.array_equal(O.resize_bilinear_u8(a, S, S), Z[f"resize.{seed}.out"])


@pytest.mark.parametrize("seed,H,W,oh,ow,C", G.NEAREST_CASES)
def test_oracle_nearest_matches_pillow_and_torch_fixtures(seed, H, W, oh, ow, C):
    m = G.nearest_source(seed, H, W)
    assert _crc(m) == int(Z[f"nearest.{seed}.crc"][0])
    assert np.array_equal(O.resize_nearest_lut(m, oh, ow, "pil"), Z[f"nearest.{seed}.pil"])
    assert np.array_equal(O.resize_nearest_lut(m, oh, ow, "torch"), Z[f"nearest.{seed}.torch"])
    lut = (np.arange(256) * 7 % C).astype(np.uint8)
    assert np.array_equal(O.resize_nearest_lut(m, oh, ow, "pil", lut), lut[Z[f"nearest.{seed}.pil"]])


def _metric_case(seed, gs, ps, C):
    gt, pred = G.metric_pair(seed, gs, ps, C, lambda g: O.resize_nearest_lut(g, ps, ps, "pil"))
    assert _crc(gt) ^ _crc(pred) == int(Z[f"metric.{seed}.crc"][0])
    return gt, pred


def _check_metrics(m, seed):
    exp = Z[f"metric.{seed}.scalars"]
    got = np.array([m["Accuracy"], m["Mean_IoU"], m["Mean_Dice"]])
    assert np.array_equal(got, exp, equal_nan=True), (got, exp)   # same integer counts, same formulas: identical doubles
    assert m["GT_Classes"] == list(Z[f"metric.{seed}.gt_classes"])
    assert m["Pred_Classes"] == list(Z[f"metric.{seed}.pred_classes"])
    assert m["Missing_Classes"] == list(Z[f"metric.{seed}.missing_classes"])
    assert m["False_Positive_Classes"] == list(Z[f"metric.{seed}.false_positive_classes"])


@pytest.mark.parametrize("seed,gs,ps,C", G.METRIC_CASES)
def test_oracle_metrics_match_reference_statements(seed, gs, ps, C):
    gt, pred = _metric_case(seed, gs, ps, C)
    _check_metrics(O.image_metrics(pred, gt, C), seed)


@pytest.mark.parametrize("seed,gs,ps,C", G.METRIC_CASES)
def test_host_metric_formulas_on_integer_counts(seed, gs, ps, C):
    """visiontransformer_amd.metrics.metrics_from_counts (what runs after the GPU counting kernel) fed with counts
    made by numpy: same CSV columns as the reference's loop, bit for bit."""
    from visiontransformer_amd.metrics import CSV_COLUMNS, csv_row, metrics_from_counts
    gt, pred = _metric_case(seed, gs, ps, C)
    counts = O.class_counts(pred, O.resize_nearest_lut(gt, ps, ps, "pil"), 256)
    m = metrics_from_counts(counts, C, pred.size)
    _check_metrics(m, seed)
    assert np.array_equal(np.array(m["ious"]), Z[f"metric.{seed}.ious"], equal_nan=True)
    assert np.array_equal(np.array(m["dices"]), Z[f"metric.{seed}.dices"], equal_nan=True)
    row = csv_row((0, "ViT", 16, 768, 12, 12), 3, 1, m, 0.01)
    assert len(row) == len(CSV_COLUMNS) and row[12] == "|".join(map(str, m["GT_Classes"]))


@pytest.mark.parametrize("i,o", [(53, 64), (160, 96), (800, 224), (255, 256), (4032, 512), (224, 224), (17, 32), (7, 512)])
def test_library_resize_tables_match_oracle(i, o):
    """vitseg_resize_coeffs / vitseg_nearest_index are host functions of libvitseg (no GPU needed)."""
    from visiontransformer_amd import preprocess as P
    taps, b, k = P.resize_tables(i, o)
    t2, b2, k2 = O.bilinear_coeffs(i, o)
    assert taps == t2 and np.array_equal(b, b2) and np.array_equal(k, k2)
    for mode, name in ((P.NEAREST_PIL, "pil"), (P.NEAREST_TORCH, "torch")):
        assert np.array_equal(P.nearest_table(i, o, mode), O.nearest_index(o, i, name))


def test_oracle_against_live_pillow_random_sizes():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(7)
    for _ in range(12):
        H, W, oh, ow = (int(v) for v in rs.randint(5, 400, size=4))
        a = rs.randint(0, 256, size=(H, W, 3), dtype=np.uint8)
        assert np.array_equal(O.resize_bilinear_u8(a, oh, ow), np.array(Image.fromarray(a, "RGB").resize((ow, oh), Image.BILINEAR)))
        m = a[:, :, 0]
        assert np.array_equal(O.resize_nearest_lut(m, oh, ow, "pil"), np.array(Image.fromarray(m).resize((ow, oh), Image.NEAREST)))
    x = O.preprocess_image(a, 32)
    assert x.dtype == np.float32 and x.shape == (3, 32, 32) and 0.0 <= x.min() and x.max() <= 1.0
