"""ORACLE tooling -- golden values of the PAED loss tails from the REAL reference functions.

`paed_loss_multiclass_soft` (model/PAED/classes.py:336-369) and the `PAEDTrainer` methods `dice_loss`
(:608-620) and `paed_loss_soft` (:623-661) use nothing but torch, so they are AST-extracted from the
reference file and evaluated on seeded tensors; inputs are regenerated from the seed by the tests.

    python oracle/make_golden_paed.py        # writes tests/golden/paed_losses.npz
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/model/PAED/classes.py"


def paed_inputs(seed=0, B=2, C=17, S=48):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, C, S, S, generator=g)
    y = torch.randint(0, C, (B, S, S), generator=g)
    blogits = torch.randn(B, 1, S, S, generator=g)
    bmask = (torch.rand(B, 1, S, S, generator=g) > 0.6).float()
    sdf_ext = torch.rand(B, 1, 32, 32, generator=g) * 5
    sdf_int = torch.rand(B, 1, 32, 32, generator=g) * 3
    return logits, y, blogits, bmask, sdf_ext, sdf_int


def main():
    tree = ast.parse(open(REF).read())
    ns = dict(torch=torch, F=F)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "paed_loss_multiclass_soft"][0]
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "PAEDTrainer"][0]
    meths = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("dice_loss", "paed_loss_soft")]
    exec(compile(ast.Module(body=[fn] + meths, type_ignores=[]), REF, "exec"), ns)
    logits, y, blogits, bmask, sdf_ext, sdf_int = paed_inputs()
    out = {}
    lg = logits.clone().requires_grad_(True)
    probs = torch.softmax(lg, dim=1)
    onehot = F.one_hot(y, 17).permute(0, 3, 1, 2).float()
    l1 = ns["paed_loss_multiclass_soft"](onehot, probs, num_classes=17)
    l1.backward()
    out["multiclass.loss"] = np.array([l1.item()])
    out["multiclass.grad"] = lg.grad.numpy()
    bl = blogits.clone().requires_grad_(True)
    preds = torch.sigmoid(bl)
    paed = ns["paed_loss_soft"](None, sdf_ext, sdf_int, preds)
    dice = ns["dice_loss"](None, preds, bmask)
    total = F.binary_cross_entropy(preds, bmask) + 0.1 * dice + 5.0 * torch.abs(paed)  # classes.py:679-681
    total.backward()
    out["binary.paed"] = np.array([paed.item()])
    out["binary.dice"] = np.array([dice.item()])
    out["binary.total"] = np.array([total.item()])
    out["binary.grad"] = bl.grad.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "paed", "paed_losses.npz"), **out)
    print({k: (v.ravel()[:1], v.shape) for k, v in out.items()})


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden", "paed"), exist_ok=True)
    main()
