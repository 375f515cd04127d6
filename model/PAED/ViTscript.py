#!/usr/bin/env python3
"""Binary PAED training -- MI355X counterpart of the reference's model/PAED/ViTscript.py:59-84:
PAEDTrainer(num_classes=1, P8, H1024, L16, A16) (:66), AdamW(1e-4) + ReduceLROnPlateau, accumulate_grad_batches=4,
EarlyStopping(val_loss, patience 6), fit, then validate and test.  The loss tail (sigmoid + BCE + 0.1 Dice +
5 |soft PAED|) and its gradient are libvitseg kernels (csrc/paed_binary.hip).  Synthetic (image, mask, sdf_ext, sdf_int)
batches unless --data (torch.save({"images", "masks", "sdf_ext", "sdf_int"})).

    python model/PAED/ViTscript.py --epochs 2 --batches 3 [--hidden-size 512 --layers 8 --heads 8 --patch-size 16]
"""
import argparse

import torch

from classes import PAEDTrainer
from visiontransformer_amd import dist as vdist, scripts, trainer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch-size", type=int, default=8)
    ap.add_argument("--hidden-size", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--heads", type=int, default=16)
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
    model = PAEDTrainer(1, a.patch_size, a.hidden_size, a.layers, a.heads, image_size=a.image_size, precision=a.precision,
                        device=dev)
    batches = scripts.paed_binary_batches(model.model.cfg, a.batches * a.batch_size, a.batch_size, a.data, seed=rank)
    log_dir = f"logs/vit-model/version_{a.version}"
    trainer.fit(model, batches, batches, max_epochs=a.epochs, accumulate_grad_batches=4, patience=6,
                ckpt_dir=log_dir + "/checkpoints", log_dir=log_dir, device=dev)
    if rank == 0:
        print("validate:", scripts.run_validation(model, batches, dev))
        print("test:", scripts.run_validation(model, batches, dev))


if __name__ == "__main__":
    main()
