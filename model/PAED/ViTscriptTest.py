#!/usr/bin/env python3
"""PAED evaluation -- MI355X counterpart of the reference's model/PAED/ViTscriptTest.py:110-227: per configuration ID,
PAEDTrainer(P16, H512, L8, A8) (:126), latest checkpoint of logs/vit-model/version_<ID>, `num_batches` test batches:
model.eval(), logits.sigmoid(), argmax over the class dim (:184-193), per-image accuracy / IoU / Dice / class sets and
the time per image into test/<model>/<model>_metrics.csv.

    python model/PAED/ViTscriptTest.py --ids 0 --num-batches 3
"""
import argparse
import os

import torch

from classes import PAEDTrainer
from visiontransformer_amd import scripts
from visiontransformer_amd.predict import CONFIGURATIONS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", type=int, nargs="*", default=[0])
    ap.add_argument("--num-classes", type=int, default=1)       # ViTscript.py:27
    ap.add_argument("--patch-size", type=int, default=16)
    ap.add_argument("--hidden-size", type=int, default=512)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--num-batches", type=int, default=10)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--data")
    ap.add_argument("--out", default="test")
    a = ap.parse_args()
    dev = "cuda:0"
    cwd = os.getcwd()
    for vid in a.ids:
        P, D, L, A = CONFIGURATIONS.get(vid, (a.patch_size, a.hidden_size, a.layers, a.heads))
        print(f"Testing Version {vid}: Patch Size {P}, Hidden Size {D}, Hidden Layers {L}, Attention Heads {A}")
        # the reference builds the SAME architecture for every ID here (:126) and names the outputs after the grid entry
        model = PAEDTrainer(a.num_classes, a.patch_size, a.hidden_size, a.layers, a.heads, image_size=a.image_size,
                            precision=a.precision, device=dev)
        ck = scripts.get_latest_checkpoint(vid, cwd)
        if ck:
            model.load_state_dict(torch.load(ck, map_location="cpu")["state_dict"])
        name = f"ID{vid}P{P}H{D}A{A}"
        batches = scripts.paed_binary_batches(model.model.cfg, a.num_batches * a.batch_size, a.batch_size, a.data, seed=5)
        rows = scripts.evaluate_to_csv(model, batches, (vid, name, P, D, L, A), os.path.join(cwd, a.out, name, f"{name}_metrics.csv"),
                                       max(a.num_classes, 2), a.num_batches, dev)
        print(f"{name}: {len(rows)} images evaluated -> {os.path.join(a.out, name)}")


if __name__ == "__main__":
    main()
