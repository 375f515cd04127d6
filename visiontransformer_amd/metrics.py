"""Evaluation metrics on the device.

The reference's evaluation script (model/CE/datasetTestViTmodel.py:152-227) pulls every prediction to the host and
loops over classes in numpy to get, per image: pixel accuracy (%), mean IoU and mean Dice over the classes present
(nan-aware), and the sets of ground-truth / predicted / missing / false-positive classes, written to
`<model>_metrics.csv`.  Here one kernel (`vitseg_eval_counts`) reduces each (prediction, ground truth) pair to
integer class statistics on the GPU -- the ground truth is nearest-resized on the fly exactly as
`Image.fromarray(gt).resize(pred.shape[::-1], Image.NEAREST)` does -- and the metrics follow from those integers with
the reference's own formulas, so the CSV is identical (tests/test_preproc_cpu.py, tests/test_gpu_preproc.py).
"""
from __future__ import annotations

import csv
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .preprocess import NEAREST_PIL, nearest_table

# header of <model>_metrics.csv (datasetTestViTmodel.py:166-171); compareModels.py:27-47 reads these columns
CSV_COLUMNS = ["Model_ID", "Model_Name", "Patch_Size", "Hidden_Size", "Layers", "Heads", "Batch_Num", "Image_Idx",
               "Accuracy", "Mean_IoU", "Mean_Dice", "Inference_Time", "GT_Classes", "Pred_Classes", "Missing_Classes",
               "False_Positive_Classes"]


def metrics_from_counts(counts: np.ndarray, num_classes: int, total_pixels: int) -> dict:
    """counts: int64 [3, 256] of one image (|gt & pred|, |gt|, |pred| per label value) -> the metric columns of one
    CSV row, with the reference's arithmetic (datasetTestViTmodel.py:193-219)."""
    inter, ngt, npr = (counts[i] for i in range(3))
    mism = int(total_pixels - int(inter.sum()))
    acc = 100 * (1 - mism / total_pixels)
    ious, dices = [], []
    for c in range(num_classes):
        i, union, size = inter[c], ngt[c] + npr[c] - inter[c], ngt[c] + npr[c]
        ious.append(float("nan") if union == 0 else i / union)
        dices.append(float("nan") if size == 0 else 2 * i / size)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # all-nan (no class present) -> nan, as in the reference
        miou, mdice = float(np.nanmean(ious)), float(np.nanmean(dices))
    gtc = [int(v) for v in np.nonzero(ngt)[0]]
    prc = [int(v) for v in np.nonzero(npr)[0]]
    return dict(Accuracy=acc, Mean_IoU=miou, Mean_Dice=mdice, GT_Classes=gtc, Pred_Classes=prc,
                Missing_Classes=sorted(set(gtc) - set(prc)), False_Positive_Classes=sorted(set(prc) - set(gtc)),
                ious=ious, dices=dices)


class Evaluator:
    """Per-image metrics of uint8 predictions [n, S, S] against label maps of any size, counted on the GPU."""

    def __init__(self, num_classes: int, device="cuda:0"):
        self.num_classes = int(num_classes)
        self.device = torch.device(device)
        self._near: Dict[tuple, torch.Tensor] = {}

    def _nearest(self, in_size: int, out_size: int) -> torch.Tensor:
        key = (in_size, out_size)
        if key not in self._near:
            self._near[key] = torch.from_numpy(nearest_table(in_size, out_size, NEAREST_PIL)).to(self.device)
        return self._near[key]

    def counts(self, pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
        """int64 [n, 3, 256] class statistics (device tensor)."""
        if pred.dtype != torch.uint8 or pred.dim() != 3 or pred.shape[1] != pred.shape[2]:
            raise ValueError(f"pred must be uint8 [n, S, S], got {pred.dtype} {tuple(pred.shape)}")
        if gt.dim() != 3 or gt.shape[0] != pred.shape[0]:
            raise ValueError("Number of images and masks must be equal!")  # the reference's dataset check, classes.py:34-35
        pred = pred.to(self.device).contiguous()
        gt = gt.to(self.device).to(torch.uint8).contiguous()   # astype(np.uint8) in the reference (:195)
        n, S, _ = pred.shape
        Hg, Wg = gt.shape[1:]
        same = (Hg, Wg) == (S, S)
        yi = None if same else self._nearest(Hg, S)
        xi = None if same else self._nearest(Wg, S)
        out = torch.empty((n, 3, 256), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().vitseg_eval_counts(pred.data_ptr(), gt.data_ptr(), n, S, Hg, Wg,
                                                     None if yi is None else yi.data_ptr(),
                                                     None if xi is None else xi.data_ptr(), out.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream))
        return out

    def evaluate(self, pred: torch.Tensor, gt: torch.Tensor) -> List[dict]:
        c = self.counts(pred, gt).cpu().numpy()
        px = int(pred.shape[1] * pred.shape[2])
        return [metrics_from_counts(c[i], self.num_classes, px) for i in range(c.shape[0])]


def csv_row(model_info: Sequence, batch_num: int, image_idx: int, m: dict, inference_time: float) -> list:
    """One row in the reference's schema; `model_info` = (Model_ID, Model_Name, Patch_Size, Hidden_Size, Layers, Heads)."""
    j = lambda v: "|".join(map(str, v))
    return list(model_info) + [batch_num, image_idx, m["Accuracy"], m["Mean_IoU"], m["Mean_Dice"], inference_time,
                               j(m["GT_Classes"]), j(m["Pred_Classes"]), j(m["Missing_Classes"]),
                               j(m["False_Positive_Classes"])]


def write_metrics_csv(path: str, rows: Sequence[Sequence]) -> None:
    with open(path, mode="w", newline="") as f:
        w = csv.writer(f)
        w.writerow(CSV_COLUMNS)
        w.writerows(rows)
