#!/usr/bin/env python3
"""Generates tests/golden/preproc/*.npz: outputs of the REAL dependencies the reference's pre-/post-processing runs on
(Pillow 12.2.0 for Resize, torch for F.interpolate) and of the reference's own metric statements, AST-extracted from
model/CE/datasetTestViTmodel.py and executed here.  Run in the build container only (needs /root/reference and PIL):

    python oracle/make_golden_preproc.py

Inputs are regenerated from seeds by the tests (np.random.RandomState, frozen legacy generator); a CRC of every input
is stored so a drifting generator is detected instead of silently comparing different data."""
import ast
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "preproc")
REF = "/root/reference/model/CE/datasetTestViTmodel.py"

# (seed, H, W, S): up-scaling, strong / anisotropic down-scaling, near-identity, identity, photo-like aspect
RESIZE_CASES = [(1, 37, 53, 64), (2, 120, 160, 96), (3, 300, 200, 128), (4, 257, 255, 256), (5, 64, 48, 224),
                (6, 600, 800, 224), (7, 224, 224, 224), (8, 17, 400, 32), (9, 756, 1008, 512)]
# (seed, H, W, out_h, out_w, classes)
NEAREST_CASES = [(11, 300, 200, 256, 256, 17), (12, 256, 256, 224, 224, 17), (13, 1000, 750, 256, 256, 4),
                 (14, 256, 256, 512, 512, 2), (15, 97, 131, 224, 224, 17)]
# (seed, gt side, pred side, classes)
METRIC_CASES = [(21, 256, 224, 17), (22, 256, 512, 2), (23, 256, 224, 5), (24, 100, 224, 17)]


def image(seed, H, W, C=3):
    return np.random.RandomState(seed).randint(0, 256, size=(H, W, C), dtype=np.uint8)


def labels(seed, H, W, classes, blocky=True):
    """Label maps with contiguous regions (so some classes are absent and IoU/Dice hit their nan branches)."""
    rs = np.random.RandomState(seed)
    coarse = rs.randint(0, max(2, classes - 3), size=((H + 15) // 16, (W + 15) // 16)).astype(np.uint8)
    return np.kron(coarse, np.ones((16, 16), np.uint8))[:H, :W] if blocky else coarse


def reference_metric_statements():
    from PIL import Image  # noqa: F401  (the extracted statements use it through the exec namespace)
    """The statements between `for idx, (image, gt_mask, pr_mask) in ...` and `writer.writerow(...)` of the reference's
    evaluation loop, plus its nested dice_coefficient function, as compiled code objects."""
    tree = ast.parse(open(REF).read())
    dice = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "dice_coefficient")
    loops = [n for n in ast.walk(tree) if isinstance(n, ast.For) and isinstance(n.target, ast.Tuple)
             and "pr_mask" in ast.dump(n.target) and any("writerow" in ast.dump(s) for s in n.body)]
    assert len(loops) == 1, "reference layout changed"
    body = [s for s in loops[0].body if "writerow" not in ast.dump(s)]
    mod = ast.Module(body=[dice] + body, type_ignores=[])
    return compile(ast.fix_missing_locations(mod), REF, "exec")


def nearest_source(seed, H, W):
    return labels(seed, H, W, 256, blocky=False) if seed == 15 else (image(seed, H, W, 1)[:, :, 0] % 23)


def metric_pair(seed, gs, ps, C, gt_to_pred_size):
    """(gt [gs, gs], pred [ps, ps]) with partial agreement; `gt_to_pred_size(gt)` = PIL-NEAREST resize to (ps, ps)."""
    gt = labels(seed, gs, gs, C)
    pred = labels(seed + 100, ps, ps, C)
    pred[: ps // 3] = gt_to_pred_size(gt)[: ps // 3]
    return gt, pred


def main():
    from PIL import Image
    os.makedirs(OUT, exist_ok=True)
    z = {}
    for seed, H, W, S in RESIZE_CASES:
        a = image(seed, H, W)
        z[f"resize.{seed}.crc"] = np.array([zlib.crc32(a.tobytes())], np.int64)
        z[f"resize.{seed}.out"] = np.array(Image.fromarray(a, "RGB").resize((S, S), Image.BILINEAR))
    for seed, H, W, oh, ow, C in NEAREST_CASES:
        m = nearest_source(seed, H, W)
        z[f"nearest.{seed}.crc"] = np.array([zlib.crc32(m.tobytes())], np.int64)
        z[f"nearest.{seed}.pil"] = np.array(Image.fromarray(m).resize((ow, oh), Image.NEAREST))
        z[f"nearest.{seed}.torch"] = F.interpolate(torch.from_numpy(m)[None, None].float(), size=(oh, ow),
                                                   mode="nearest")[0, 0].to(torch.uint8).numpy()
    code = reference_metric_statements()
    for seed, gs, ps, C in METRIC_CASES:
        gt, pred = metric_pair(seed, gs, ps, C, lambda g: np.array(Image.fromarray(g).resize((ps, ps), Image.NEAREST)))
        # pr_mask is what the loop sees: per-class scores whose argmax is `pred`
        pr_mask = torch.from_numpy(np.eye(C, dtype=np.float32)[pred].transpose(2, 0, 1).copy())
        ns = dict(np=np, Image=Image, torch=torch, num_classes=C, gt_mask=torch.from_numpy(gt.astype(np.int64)),
                  pr_mask=pr_mask, image=None)
        exec(code, ns)
        z[f"metric.{seed}.crc"] = np.array([zlib.crc32(gt.tobytes()) ^ zlib.crc32(pred.tobytes())], np.int64)
        z[f"metric.{seed}.scalars"] = np.array([ns["accuracy"], ns["mean_iou"], ns["mean_dice"]], np.float64)
        for k in ("gt_classes", "pred_classes", "missing_classes", "false_positive_classes"):
            z[f"metric.{seed}.{k}"] = np.array(ns[k], np.int64)
        z[f"metric.{seed}.ious"] = np.array(ns["ious"], np.float64)
        z[f"metric.{seed}.dices"] = np.array(ns["dices"], np.float64)
    np.savez_compressed(os.path.join(OUT, "preproc.npz"), **z)
    print("wrote", os.path.join(OUT, "preproc.npz"), os.path.getsize(os.path.join(OUT, "preproc.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
