"""Shared helpers for the parity tests (golden loading, oracle wiring)."""
import os

import numpy as np
import torch

from visiontransformer_amd import synth
from visiontransformer_amd.config import ViTSegConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        m = [int(v) for v in self.z["meta.cfg"]]
        self.cfg = ViTSegConfig(m[0], m[1], m[2], m[3], m[4], image_size=m[5], intermediate_size=m[6])
        self.batch, self.wseed = m[7], m[8]
        self.head_gain = float(self.z["meta.head_gain"][0])

    def state_dict_np(self):
        return synth.make_state_dict(self.cfg, seed=self.wseed, head_gain=self.head_gain)

    def state_dict(self, dtype=torch.float32):
        return {k: torch.from_numpy(v).to(dtype) for k, v in self.state_dict_np().items()}

    def images(self):
        return torch.from_numpy(synth.make_images(self.cfg, self.batch, seed=0))

    def targets(self):
        return torch.from_numpy(synth.make_targets(self.cfg, self.batch, seed=0))

    def has(self, key):
        return key + ".idx" in self.z.files

    def sampled(self, key):
        return self.z[key + ".idx"], self.z[key + ".val"], tuple(self.z[key + ".shape"])

    def max_abs_err(self, key, tensor):
        """max |tensor.flat[idx] - golden| and the golden's max-abs (for relative scale)."""
        idx, val, shape = self.sampled(key)
        assert tuple(tensor.shape) == shape, (key, tuple(tensor.shape), shape)
        a = tensor.detach().to(torch.float64).cpu().numpy().ravel()[idx]
        return float(np.abs(a - val.astype(np.float64)).max()), float(np.abs(val).max())

    def checksum_rel_err(self, key, tensor):
        a = tensor.detach().to(torch.float64).cpu().numpy().ravel()
        s = self.z[key + ".sum"]
        return abs(a.sum() - s[0]) / (np.sqrt(s[1] * a.size) + 1e-30)

    def mask(self):
        shape = tuple(self.z["mask.shape"])
        if "mask.bits" in self.z.files:
            n = int(np.prod(shape))
            return np.unpackbits(self.z["mask.bits"])[:n].reshape(shape)
        return self.z["mask.u8"]

    def fragile(self):
        shape = tuple(self.z["mask.shape"])
        n = int(np.prod(shape))
        return np.unpackbits(self.z["mask.fragile_bits"])[:n].reshape(shape).astype(bool)
