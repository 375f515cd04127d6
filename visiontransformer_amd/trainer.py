"""Minimal trainer: what the reference gets from `L.Trainer(...).fit()` in its train scripts
(model/CE/createViTmodel.py:62-78, model/CE/trainCurrentViTmodel.py:63-73), without Lightning:
gradient accumulation (accumulate_grad_batches=4), EarlyStopping on `valid_loss`, ModelCheckpoint files in
Lightning's layout (`{'state_dict': ...}` with `model.`-prefixed keys, named `epoch=E-step=S.ckpt` so the
reference's `get_latest_checkpoint`, datasetTestViTmodel.py:38-54, finds them) and a `metrics.csv` with the
CSVLogger column names.  Data-parallel when launched under torchrun: one process per GPU; gradients are
accumulated locally over the micro-batches of an optimizer step and summed over RCCL ONCE, on its last micro-batch
(`model.no_sync()` around the others, as DDP's no_sync).  Checkpoints carry the optimizer state (Adam moments and step
count, Lightning's `optimizer_states`), so a resumed run continues with the update the uninterrupted run would have made.
No per-step host synchronisation: losses stay device scalars until the epoch's row is written.
"""
from __future__ import annotations

import contextlib
import csv
import inspect
import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist

from .dist import sync_grads
from .lightning import LightningViTModel


def fit(model, train_batches: Iterable, val_batches: Optional[Iterable] = None, *,
        max_epochs: int = 100, accumulate_grad_batches: int = 4, patience: int = 3, ckpt_dir: Optional[str] = None,
        log_dir: Optional[str] = None, resume_from: Optional[str] = None, device="cuda:0"):
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    opt = model.configure_optimizers()
    sched = None
    if isinstance(opt, dict):                      # Lightning's dict form (PAEDTrainer: AdamW + ReduceLROnPlateau)
        sched = opt.get("lr_scheduler")
        opt = opt["optimizer"]
    from .optim import FusedAdam
    fused = isinstance(opt, FusedAdam)             # its step() folds the 1 / world of data parallelism into the kernel
    net = getattr(model, "model", model)           # the ViTSegmentationModel that owns the arena
    start_epoch, step = 0, 0
    if resume_from:
        ck = torch.load(resume_from, map_location="cpu")
        model.load_state_dict(ck["state_dict"])
        if ck.get("optimizer_states"):             # trainer.fit(ckpt_path=...) restores these in the reference's runs
            opt.load_state_dict(ck["optimizer_states"][0])
            for st in opt.state.values():          # moments live with the parameters
                for k, v in st.items():
                    if torch.is_tensor(v):
                        st[k] = v.to(device)
        start_epoch, step = int(ck.get("epoch", -1)) + 1, int(ck.get("global_step", 0))
    rows, best, bad = [], float("inf"), 0
    takes_scale = "grad_scale" in inspect.signature(model.training_step).parameters
    for epoch in range(start_epoch, max_epochs):
        model.train()
        ep_loss, n_micro = None, 0
        opt.zero_grad(set_to_none=True)
        pending = []
        for i, batch in enumerate(train_batches):
            batch = tuple(t.to(device) for t in batch)
            # Lightning also steps on the epoch's final batch when the accumulation window is incomplete
            n_total = len(train_batches) if hasattr(train_batches, "__len__") else None
            last = (n_micro + 1) % accumulate_grad_batches == 0 or (n_total is not None and i + 1 == n_total)
            # every micro-batch loss is scaled by 1 / accumulate (Lightning does the same); all but the last backward of
            # an optimizer step stay local
            with (contextlib.nullcontext() if last else net.no_sync()):
                if takes_scale:
                    loss = model.training_step(batch, i, grad_scale=1.0 / accumulate_grad_batches)
                    loss.backward()
                else:
                    loss = model.training_step(batch, i)
                    (loss / accumulate_grad_batches).backward()
            ld = loss.detach()
            ep_loss = ld if ep_loss is None else ep_loss + ld
            n_micro += 1
            if last:
                sync_grads(net)
                if fused:
                    opt.step(grad_scale=1.0 / world)
                else:
                    if world > 1:
                        net.arena.grad.mul_(1.0 / world)
                    opt.step()
                opt.zero_grad(set_to_none=True)
                step += 1
                pending.append(dict(epoch=epoch, step=step, train_loss_step=ld))
        for r in pending:                              # one host read per logged value, after the epoch's last kernel
            r["train_loss_step"] = float(r["train_loss_step"])
        rows.extend(pending)
        row = dict(epoch=epoch, step=step, train_loss_epoch=(float(ep_loss) / n_micro if n_micro else 0.0))
        if val_batches is not None:
            model.eval()
            vs = [model.validation_step(tuple(t.to(device) for t in batch), i) for i, batch in enumerate(val_batches)]
            v = float(torch.stack([t.detach().float().reshape(()) for t in vs]).mean()) if vs else 0.0
            if world > 1:
                t = torch.tensor([v], device=device)
                dist.all_reduce(t)
                v = float(t) / world
            row["valid_loss"] = v
            if sched is not None:
                sched["scheduler"].step(v)
        rows.append(row)
        if rank == 0 and ckpt_dir:
            os.makedirs(ckpt_dir, exist_ok=True)
            osd = opt.state_dict()
            osd = {"state": {k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in st.items()}
                             for k, st in osd["state"].items()}, "param_groups": osd["param_groups"]}
            torch.save({"state_dict": {k: t.cpu() for k, t in model.state_dict().items()}, "epoch": epoch,
                        "global_step": step, "optimizer_states": [osd]},
                       os.path.join(ckpt_dir, f"epoch={epoch}-step={step}.ckpt"))
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
