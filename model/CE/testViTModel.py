#!/usr/bin/env python3
"""Single-image inference, the MI355X counterpart of the reference's model/CE/testViTModel.py:92-126.

    python model/CE/testViTModel.py IMAGE [--model-id 0] [--num-classes 17] [--checkpoint x.ckpt]
                                          [--image-size 224] [--precision fp32|bf16] [--out mask.png]

Without --checkpoint the weights are random (the reference's checkpoints are private)."""
import argparse

import numpy as np

from classes import LightningViTModel  # noqa: F401  (same import the reference script uses)
from visiontransformer_amd.predict import load_model, predict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("image")
    ap.add_argument("--model-id", type=int, default=0, help="configuration ID 0..8 of the reference grid")
    ap.add_argument("--num-classes", type=int, default=17)
    ap.add_argument("--checkpoint")
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--out")
    a = ap.parse_args()
    model = load_model(a.model_id, a.num_classes, a.checkpoint, image_size=a.image_size, precision=a.precision)
    mask = predict(a.image, model)
    print("classes present:", np.unique(mask).tolist(), "mask shape:", mask.shape)
    if a.out:
        from PIL import Image
        Image.fromarray((mask.astype(np.float32) * (255.0 / max(a.num_classes - 1, 1))).astype(np.uint8)).save(a.out)


if __name__ == "__main__":
    main()
