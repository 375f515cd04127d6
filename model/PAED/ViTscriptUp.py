#!/usr/bin/env python3
"""Resumed 17-class soft-PAED training -- MI355X counterpart of the reference's model/PAED/ViTscriptUp.py:62-90:
LightningViTModel(P4, H1024, L16, A16) (:62; its class count is forced to 17), trainer.fit(..., ckpt_path=<checkpoint>)
with EarlyStopping(val_loss, patience 5).  The checkpoint restores weights AND optimizer state (Adam moments, step).

    python model/PAED/ViTscriptUp.py --version 0 [--resume logs/vit-model/version_0/checkpoints/epoch=0-step=2.ckpt]
"""
import argparse
import os

import torch

from classes import LightningViTModel
from visiontransformer_amd import dist as vdist, scripts, trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch-size", type=int, default=4)
    ap.add_argument("--hidden-size", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--version", type=int, default=0)
    ap.add_argument("--resume", help="checkpoint to continue from (default: the latest of --version, if any)")
    ap.add_argument("--data")
    a = ap.parse_args()
    rank, world, local = vdist.init()
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    model = LightningViTModel(17, a.patch_size, a.hidden_size, a.layers, a.heads, image_size=a.image_size,
                              precision=a.precision, device=dev)
    batches = scripts.ce_batches(model.model.cfg, a.batches * a.batch_size, a.batch_size, a.data, first=rank * a.batches * a.batch_size)
    log_dir = f"logs/vit-model/version_{a.version}"
    resume = a.resume or scripts.get_latest_checkpoint(a.version, os.getcwd())
    rows = trainer.fit(model, batches, batches, max_epochs=a.epochs, accumulate_grad_batches=1, patience=5,
                       ckpt_dir=log_dir + "/checkpoints", log_dir=log_dir, resume_from=resume, device=dev)
    if rank == 0:
        print(rows[-1] if rows else "nothing left to train")
        print("validate:", scripts.run_validation(model, batches, dev))


if __name__ == "__main__":
    main()
