#!/usr/bin/env python3
"""CE training from scratch -- MI355X counterpart of the reference's model/CE/createViTmodel.py:53-78:
LightningViTModel(P16, H1024, L16, A16), Adam(lr=1e-5), accumulate_grad_batches=4, EarlyStopping(valid_loss, patience 3),
CSV log + checkpoints under logs/vit-model/version_<n>, then validate and test (the reference reuses one folder for all
three splits, :40-47).  Synthetic batches unless --data (torch.save({"images", "masks"})).

    python model/CE/createViTmodel.py --epochs 2 --batches 3 [--version 0] [--precision bf16]
"""
import argparse

import torch

from classes import LightningViTModel
from visiontransformer_amd import dist as vdist, scripts, trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch-size", type=int, default=16)
    ap.add_argument("--hidden-size", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--num-classes", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--version", type=int, default=0)
    ap.add_argument("--data")
    a = ap.parse_args()
    rank, world, local = vdist.init()
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    model = LightningViTModel(a.num_classes, a.patch_size, a.hidden_size, a.layers, a.heads, image_size=a.image_size,
                              precision=a.precision, device=dev)
    batches = scripts.ce_batches(model.model.cfg, a.batches * a.batch_size, a.batch_size, a.data, first=rank * a.batches * a.batch_size)
    log_dir = f"logs/vit-model/version_{a.version}"
    trainer.fit(model, batches, batches, max_epochs=a.epochs, accumulate_grad_batches=4, patience=3,
                ckpt_dir=log_dir + "/checkpoints", log_dir=log_dir, device=dev)
    if rank == 0:
        print("validate:", scripts.run_validation(model, batches, dev))
        print("test:", scripts.run_validation(model, batches, dev))


if __name__ == "__main__":
    main()
