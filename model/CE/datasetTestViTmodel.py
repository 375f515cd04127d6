#!/usr/bin/env python3
"""Dataset evaluation -- MI355X counterpart of the reference's model/CE/datasetTestViTmodel.py:107-227: for every
configuration ID, build LightningViTModel, load the latest checkpoint of logs/vit-model/version_<ID> (the reference
resumes the Lightning trainer on it; only the weights matter for the evaluation), run `num_batches` test batches and
write test/<model>/<model>_metrics.csv (per-image accuracy / mean IoU / mean Dice / class sets / time per image) --
the schema model/CE/compareModels.py:27-47 reads.  Plots are host-side reporting and not produced.

    python model/CE/datasetTestViTmodel.py --ids 1 --num-classes 17 --num-batches 3 [--data eval.pt]
"""
import argparse
import os

import torch

from classes import LightningViTModel
from visiontransformer_amd import scripts
from visiontransformer_amd.predict import CONFIGURATIONS


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ids", type=int, nargs="*", default=sorted(CONFIGURATIONS))
    ap.add_argument("--num-classes", type=int, default=17)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=4)          # DataLoader(batch_size=4), :103
    ap.add_argument("--num-batches", type=int, default=10)        # :151
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--data")
    ap.add_argument("--out", default="test")
    a = ap.parse_args()
    dev = "cuda:0"
    cwd = os.getcwd()
    for vid in a.ids:
        P, D, L, A = CONFIGURATIONS[vid]
        print(f"Testing Version {vid}: Patch Size {P}, Hidden Size {D}, Hidden Layers {L}, Attention Heads {A}")
        model = LightningViTModel(a.num_classes, P, D, L, A, image_size=a.image_size, precision=a.precision, device=dev)
        ck = scripts.get_latest_checkpoint(vid, cwd)
        if ck:
            model.load_state_dict(torch.load(ck, map_location="cpu")["state_dict"])
        name = f"ID{vid}P{P}H{D}A{A}"
        batches = scripts.ce_batches(model.model.cfg, a.num_batches * a.batch_size, a.batch_size, a.data, seed=3)
        rows = scripts.evaluate_to_csv(model, batches, (vid, name, P, D, L, A), os.path.join(cwd, a.out, name, f"{name}_metrics.csv"),
                                       a.num_classes, a.num_batches, dev)
        acc = sum(r[8] for r in rows) / max(len(rows), 1)
        print(f"{name}: {len(rows)} images, mean accuracy {acc:.2f} %, {rows[0][11] * 1e3:.2f} ms/image")


if __name__ == "__main__":
    main()
