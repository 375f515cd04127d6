"""What the reference's train / evaluation scripts share (model/CE/*.py, model/PAED/*.py), for the thin drivers under
model/: batches (the reference's dataset VisionChallenge/ is private, so synthetic tensors unless --data points at a
torch.save'd dict), checkpoint discovery with the reference's rule, and the per-image evaluation loop + CSV.
"""
from __future__ import annotations

import os
import re
import time
from typing import List, Optional

import numpy as np
import torch

from . import synth
from .metrics import Evaluator, csv_row, write_metrics_csv


def checkpoint_epoch(path: str) -> Optional[int]:
    m = re.search(r"epoch=(\d+)", path)
    return int(m.group(1)) if m else None


def get_latest_checkpoint(version_n: int, base_path: str) -> Optional[str]:
    """The checkpoint with the largest epoch number among logs/vit-model/version_<n>/checkpoints/epoch=E-step=S.ckpt,
    None (with a message) when the directory or any checkpoint is missing -- the rule of the reference's helper
    (model/CE/datasetTestViTmodel.py:38-54, also used by model/PAED/ViTscriptTest.py)."""
    ckpt_dir = os.path.join(base_path, "logs", "vit-model", f"version_{version_n}", "checkpoints")
    if not os.path.isdir(ckpt_dir):
        print(f"no checkpoint directory: {ckpt_dir}")
        return None
    by_epoch = {}
    for name in sorted(os.listdir(ckpt_dir)):
        epoch = checkpoint_epoch(name) if name.endswith(".ckpt") else None
        if epoch is not None:
            by_epoch[epoch] = name
    if not by_epoch:
        print(f"no epoch=<E>-step=<S>.ckpt file in {ckpt_dir}")
        return None
    path = os.path.join(ckpt_dir, by_epoch[max(by_epoch)])
    print(f"latest checkpoint: {path}")
    return path


def ce_batches(cfg, n_images: int, batch_size: int, data: Optional[str] = None, seed: int = 0, first: int = 0):
    """[(images [b,3,S,S] float, masks [b,256,256] long)]: StructuralDamageDataset items (model/CE/classes.py:60-89)."""
    if data:
        blob = torch.load(data)
        xs, ys = blob["images"].float(), blob["masks"].long()
    else:
        xs = torch.from_numpy(synth.make_images(cfg, n_images, seed=seed, first_image=first))
        ys = torch.from_numpy(synth.make_targets(cfg, n_images, seed=seed, first_image=first))
    return [(xs[i:i + batch_size], ys[i:i + batch_size]) for i in range(0, xs.shape[0], batch_size)]


def paed_binary_batches(cfg, n_images: int, batch_size: int, data: Optional[str] = None, seed: int = 0, sdf_size: int = 224):
    """[(images, masks [b,1,h,w] float 0/1, sdf_ext [b,h,w], sdf_int [b,h,w])]: the binary PAED dataset's items
    (model/PAED/classes.py:60-88: mask resized to 224 NEAREST and binarised, SDFs computed from it on the host)."""
    if data:
        blob = torch.load(data)
        xs, ms, se, si = blob["images"].float(), blob["masks"].float(), blob["sdf_ext"].float(), blob["sdf_int"].float()
    else:
        xs = torch.from_numpy(synth.make_images(cfg, n_images, seed=seed))
        g = torch.Generator().manual_seed(seed + 17)
        # blobs: threshold a smooth random field; the "SDFs" are smooth non-negative maps of the same size (synthetic
        # stand-ins: the real ones come from scipy distance transforms in the dataset class, host-side I/O)
        field = torch.nn.functional.avg_pool2d(torch.rand(n_images, 1, sdf_size + 30, sdf_size + 30, generator=g), 31, 1)
        ms = (field > field.mean()).float()
        se = (field[:, 0] - field.amin()).clamp_min(0) * 40 * (1 - ms[:, 0])
        si = (field.amax() - field[:, 0]).clamp_min(0) * 40 * ms[:, 0]
    out = []
    for i in range(0, xs.shape[0], batch_size):
        out.append((xs[i:i + batch_size], ms[i:i + batch_size], se[i:i + batch_size], si[i:i + batch_size]))
    return out


def run_validation(model, batches, device) -> dict:
    """trainer.validate / trainer.test of the reference scripts: validation_step over the loader, mean of what it logs."""
    model.eval()
    acc: dict = {}
    for i, batch in enumerate(batches):
        model.validation_step(tuple(t.to(device) for t in batch), i)
        for k, v in model.logged.items():
            if k.startswith("val"):
                acc.setdefault(k, []).append(v.detach().float().reshape(()) if torch.is_tensor(v) else torch.tensor(float(v)))
    return {k: float(torch.stack([t.to("cpu") for t in v]).mean()) for k, v in acc.items()}


def evaluate_to_csv(model, batches, model_info, csv_path: str, num_classes: int, num_batches: int, device) -> List[list]:
    """The per-image loop of datasetTestViTmodel.py:163-227 / ViTscriptTest.py:160-227: model.eval(), logits.sigmoid(),
    argmax over the class dim, ground truth NEAREST-resized to the prediction, accuracy / mean IoU / mean Dice / class
    sets per image, one CSV row each with the batch's average time per image.  Predictions and class statistics stay on
    the GPU (fused sigmoid -> argmax mask, vitseg_eval_counts)."""
    seg = getattr(model, "model", model)
    ev = Evaluator(num_classes, device)
    rows = []
    model.eval()
    for bn, batch in enumerate(batches):
        if bn >= num_batches:
            break
        x, gt = batch[0].to(device), batch[1]
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            mask = seg.predict_mask(x)
        torch.cuda.synchronize()
        per_image = (time.time() - t0) / len(x)
        gt = gt.reshape(gt.shape[0], gt.shape[-2], gt.shape[-1])
        for idx, m in enumerate(ev.evaluate(mask, gt)):
            rows.append(csv_row(model_info, bn, idx, m, per_image))
    os.makedirs(os.path.dirname(os.path.abspath(csv_path)), exist_ok=True)
    write_metrics_csv(csv_path, rows)
    return rows
