"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's pre- and post-processing around the hot path
(SURVEY.md section 8(f) rows f3 and f4).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product path (visiontransformer_amd/) never does.

f3  image side : `transforms.Resize((S, S))` + `transforms.ToTensor()` on a PIL RGB image
                 (model/CE/trainCurrentViTmodel.py:48-51, model/CE/testViTModel.py:92-97).  torchvision's Resize on a
                 PIL image is `Image.resize((S, S), BILINEAR)`: Pillow's two-pass (horizontal, then vertical)
                 antialiased triangle-filter resampling in 8-bit fixed point.  The arithmetic lives in the third-party
                 dependency Pillow (src/libImaging/Resample.c; 12.2.0 in this image, unpinned by the reference's
                 requirements.txt); it is restated below from its published algorithm.
    mask side  : `transforms.Resize((256, 256), NEAREST)` on the 'L' mask, the value -> class-index remap and
                 (training) `F.interpolate(..., mode='nearest')` (model/CE/classes.py:76-83, 273-274).
f4  metrics    : per-image accuracy / IoU / Dice / class sets of model/CE/datasetTestViTmodel.py:152-227.

PINNED: tests/golden/preproc/*.npz hold outputs of the real Pillow (oracle/make_golden_preproc.py, run in this
container) for up-, down- and mixed scaling, and of numpy for the metrics; tests/test_preproc_cpu.py checks this file
against them (and against the live Pillow where it is importable).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2  # Resample.c: fixed-point fraction bits of the 8 bpc path


def bilinear_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1.0) and the full
    box (0, in_size).  Returns ksize, bounds int32 [out, 2] = (first source index, tap count), kk int32 [out, ksize]."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # (double)(in1 - in0) / outSize, box is float
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.empty(xmax, np.float64)
        ww = 0.0
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w[x] = 1.0 - t if t < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):   # (int)(+-0.5 + k * 2^22): C truncation towards zero
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(src: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """One resampling pass over `axis` of a uint8 [H, W, C] image: ss = 2^21 + sum(pixel * k) in int32,
    out = clip8(ss >> 22)."""
    src = np.moveaxis(src, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i in range(bounds.shape[0]):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = np.tensordot(kk[i, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR)` for uint8 [H, W, C] (Resample.c ImagingResample:
    horizontal pass over the source rows the vertical pass needs, uint8 intermediate, then the vertical pass)."""
    H, W = img.shape[:2]
    if (H, W) == (out_h, out_w):
        return img.copy()
    cur = img
    _, yb, yk = bilinear_coeffs(H, out_h)
    if W != out_w:
        _, xb, xk = bilinear_coeffs(W, out_w)
        first, last = int(yb[0, 0]), int(yb[-1, 0] + yb[-1, 1])
        cur = _pass(img[first:last], xb, xk, axis=1)
        yb = yb.copy()
        yb[:, 0] -= first
    if H != out_h:
        cur = _pass(cur, yb, yk, axis=0)
    return cur


def to_tensor(img_u8: np.ndarray) -> np.ndarray:
    """transforms.ToTensor(): uint8 HWC -> float32 CHW, value / 255 (one correctly rounded fp32 division)."""
    return (img_u8.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32)


def preprocess_image(img_u8: np.ndarray, S: int) -> np.ndarray:
    """Compose([Resize((S, S)), ToTensor()]) on an RGB uint8 image."""
    return to_tensor(resize_bilinear_u8(img_u8, S, S))


def nearest_index(out_size: int, in_size: int, mode: str) -> np.ndarray:
    """Source index of every destination sample.  'pil': Image.resize(NEAREST) samples the pixel under the destination
    centre, walking the source coordinate incrementally (Geometry.c ImagingScaleAffine); 'torch': F.interpolate(mode='nearest') uses
    floor(d * (in / out)) with the scale in fp32 (model/CE/classes.py:273-274)."""
    d = np.arange(out_size)
    if mode == "pil":
        # Geometry.c ImagingScaleAffine: xo = a0 * 0.5, then xo += a0 per destination pixel in double (the running sum
        # is NOT (d + 0.5) * a0: 256 -> 224 differs in 47 % of the pixels), index = (int)xo
        a0 = float(in_size) / out_size
        idx = np.empty(out_size, np.int64)
        xo = 0.0 + a0 * 0.5
        for i in range(out_size):
            idx[i] = int(xo)
            xo += a0
    elif mode == "torch":
        idx = np.floor((d.astype(np.float32) * np.float32(in_size / out_size))).astype(np.int64)
    else:
        raise ValueError(mode)
    return np.minimum(idx, in_size - 1)


def resize_nearest_lut(mask_u8: np.ndarray, out_h: int, out_w: int, mode: str, lut=None) -> np.ndarray:
    """Nearest resize of an 'L' mask followed by the value -> class remap (np.vectorize(value_to_class.get),
    classes.py:79-83) given as a 256-entry table."""
    r = mask_u8[nearest_index(out_h, mask_u8.shape[0], mode)][:, nearest_index(out_w, mask_u8.shape[1], mode)]
    return r if lut is None else np.asarray(lut, np.uint8)[r]


# ------------------------------------------------------------------------------------------------ f4: metrics
def class_counts(pred: np.ndarray, gt: np.ndarray, num_classes: int) -> np.ndarray:
    """int64 [3, C]: per class |gt & pred|, |gt|, |pred| of one image (gt already at the prediction's size)."""
    out = np.zeros((3, num_classes), np.int64)
    for c in range(num_classes):
        g, p = gt == c, pred == c
        out[0, c] = np.logical_and(g, p).sum()
        out[1, c] = g.sum()
        out[2, c] = p.sum()
    return out


def image_metrics(pred: np.ndarray, gt_mask: np.ndarray, num_classes: int) -> dict:
    """One CSV row's metric columns, following datasetTestViTmodel.py:193-219: the ground truth is resized to the
    prediction's shape with PIL NEAREST, accuracy in percent, nan-aware means of the per-class IoU and Dice, and the
    class sets."""
    gt = resize_nearest_lut(gt_mask.astype(np.uint8), pred.shape[0], pred.shape[1], "pil")
    mism = int((gt != pred).astype(float).sum())
    acc = 100 * (1 - mism / pred.size)
    ious, dices = [], []
    for c in range(num_classes):
        g, p = gt == c, pred == c
        inter, union = np.logical_and(g, p).sum(), np.logical_or(g, p).sum()
        ious.append(float("nan") if union == 0 else inter / union)
        size = g.sum() + p.sum()
        dices.append(float("nan") if size == 0 else 2 * inter / size)
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            miou, mdice = np.nanmean(ious), np.nanmean(dices)
    gtc = sorted(set(int(c) for c in np.unique(gt)))
    prc = sorted(set(int(c) for c in np.unique(pred)))
    return dict(Accuracy=acc, Mean_IoU=float(miou), Mean_Dice=float(mdice), GT_Classes=gtc, Pred_Classes=prc,
                Missing_Classes=sorted(set(gtc) - set(prc)), False_Positive_Classes=sorted(set(prc) - set(gtc)))
