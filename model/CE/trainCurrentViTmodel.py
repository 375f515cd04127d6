#!/usr/bin/env python3
"""CE training driver -- MI355X counterpart of the reference's model/CE/trainCurrentViTmodel.py:22-73
(and createViTmodel.py): LightningViTModel + Adam(lr=1e-5), accumulate_grad_batches=4, EarlyStopping on
valid_loss, checkpoints in Lightning's layout.  The reference's dataset (VisionChallenge/) is private,
so batches come from the procedural generator unless --data points at tensors saved with torch.save
({"images": [N,3,S,S] float, "masks": [N,256,256] long}).

    python model/CE/trainCurrentViTmodel.py --model-id 0 --num-classes 2 --image-size 224 --epochs 2
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 model/CE/trainCurrentViTmodel.py ...
"""
import argparse

import torch

from classes import LightningViTModel  # noqa: F401  (the import the reference script uses)
from visiontransformer_amd import dist as vdist, synth, trainer
from visiontransformer_amd.predict import CONFIGURATIONS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-id", type=int, default=8, help="reference configuration ID (default 8 = P4 H1024 L16, :63)")
    ap.add_argument("--num-classes", type=int, default=2)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=4)      # DataLoader(batch_size=4), :57
    ap.add_argument("--batches", type=int, default=8, help="synthetic batches per epoch")
    ap.add_argument("--epochs", type=int, default=100)         # max_epochs=100, :72
    ap.add_argument("--patience", type=int, default=3)
    ap.add_argument("--data")
    ap.add_argument("--ckpt-dir", default="logs/vit-model/version_0/checkpoints")
    ap.add_argument("--resume")
    a = ap.parse_args()
    rank, world, local = vdist.init()
    dev = f"cuda:{local}"
    torch.cuda.set_device(local)
    P, D, L, A = CONFIGURATIONS[a.model_id]
    model = LightningViTModel(a.num_classes, P, D, L, A, image_size=a.image_size, device=dev)
    cfg = model.model.cfg
    if a.data:
        blob = torch.load(a.data)
        xs, ys = blob["images"].float(), blob["masks"].long()
    else:
        n = a.batches * a.batch_size * world
        xs = torch.from_numpy(synth.make_images(cfg, n, seed=0))
        ys = torch.from_numpy(synth.make_targets(cfg, n, seed=0))
    lo, hi = vdist.shard_range(xs.shape[0], rank, world)
    xs, ys = xs[lo:hi], ys[lo:hi]
    batches = [(xs[i:i + a.batch_size], ys[i:i + a.batch_size]) for i in range(0, xs.shape[0], a.batch_size)]
    rows = trainer.fit(model, batches, batches, max_epochs=a.epochs, accumulate_grad_batches=4, patience=a.patience,
                       ckpt_dir=a.ckpt_dir, log_dir="logs/vit-model/version_0", resume_from=a.resume, device=dev)
    if rank == 0:
        print(rows[-1])


if __name__ == "__main__":
    main()
