"""Minimal trainer: what the reference gets from `L.Trainer(...).fit()` in its train scripts
(model/CE/createViTmodel.py:62-78, model/CE/trainCurrentViTmodel.py:63-73), without Lightning:
gradient accumulation (accumulate_grad_batches=4), EarlyStopping on `valid_loss`, ModelCheckpoint files in
Lightning's layout (`{'state_dict': ...}` with `model.`-prefixed keys, named `epoch=E-step=S.ckpt` so the
reference's `get_latest_checkpoint`, datasetTestViTmodel.py:38-54, finds them) and a `metrics.csv` with the
CSVLogger column names.  Data-parallel when launched under torchrun: one process per GPU, gradients summed
over RCCL after the last micro-batch of an optimizer step.
"""
from __future__ import annotations

import csv
import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from .dist import sync_grads
from .lightning import LightningViTModel


def fit(model: LightningViTModel, train_batches: Iterable, val_batches: Optional[Iterable] = None, *,
        max_epochs: int = 100, accumulate_grad_batches: int = 4, patience: int = 3, ckpt_dir: Optional[str] = None,
        log_dir: Optional[str] = None, resume_from: Optional[str] = None, device="cuda:0"):
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    opt = model.configure_optimizers()
    start_epoch, step = 0, 0
    if resume_from:
        ck = torch.load(resume_from, map_location="cpu")
        model.load_state_dict(ck["state_dict"])
        start_epoch, step = int(ck.get("epoch", -1)) + 1, int(ck.get("global_step", 0))
    rows, best, bad = [], float("inf"), 0
    for epoch in range(start_epoch, max_epochs):
        model.train()
        ep_loss, n_micro = 0.0, 0
        opt.zero_grad(set_to_none=True)
        for i, (x, y) in enumerate(train_batches):
            loss = model.training_step((x.to(device), y.to(device)), i)
            (loss / accumulate_grad_batches).backward()  # Lightning scales each micro-batch loss the same way
            ep_loss += float(loss.detach())
            n_micro += 1
            if n_micro % accumulate_grad_batches == 0:
                sync_grads(model.model)
                opt.step(grad_scale=1.0 / world)
                opt.zero_grad(set_to_none=True)
                step += 1
                rows.append(dict(epoch=epoch, step=step, train_loss_step=float(loss.detach())))
        row = dict(epoch=epoch, step=step, train_loss_epoch=ep_loss / max(n_micro, 1))
        if val_batches is not None:
            model.eval()
            vs = [float(model.validation_step((x.to(device), y.to(device)), i)) for i, (x, y) in enumerate(val_batches)]
            v = sum(vs) / max(len(vs), 1)
            if world > 1:
                t = torch.tensor([v], device=device)
                dist.all_reduce(t)
                v = float(t) / world
            row["valid_loss"] = v
        rows.append(row)
        if rank == 0 and ckpt_dir:
            os.makedirs(ckpt_dir, exist_ok=True)
            torch.save({"state_dict": {k: t.cpu() for k, t in model.state_dict().items()}, "epoch": epoch,
                        "global_step": step}, os.path.join(ckpt_dir, f"epoch={epoch}-step={step}.ckpt"))
        if rank == 0 and log_dir:
            os.makedirs(log_dir, exist_ok=True)
            cols = ["epoch", "step", "train_loss_step", "train_loss_epoch", "valid_loss"]
            with open(os.path.join(log_dir, "metrics.csv"), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=cols)
                w.writeheader()
                w.writerows(rows)
        if "valid_loss" in row:  # EarlyStopping(monitor="valid_loss", patience=...)
            if row["valid_loss"] < best - 0.0:
                best, bad = row["valid_loss"], 0
            else:
                bad += 1
                if bad >= patience:
                    break
    return rows
