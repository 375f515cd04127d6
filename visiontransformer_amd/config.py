"""Shape/config record for the ViT segmentation hot path.

Mirrors the constructor arguments of the reference model
(`/root/reference/model/CE/classes.py:222-238`): the five positional arguments
are the reference's; everything else is what the reference hard-codes
(`image_size=224`, `intermediate_size=3072`, eps 1e-12 from
`transformers/models/vit/configuration_vit.py:58`) exposed as keywords so that
BASELINE's 512x512 / 1024x1024 configurations can be expressed.
"""
from __future__ import annotations

from dataclasses import dataclass

HEAD_MID_CHANNELS = 256  # seg_head.0 out channels, reference classes.py:241


@dataclass(frozen=True)
class ViTSegConfig:
    num_classes: int
    patch_size: int
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    image_size: int = 224
    intermediate_size: int = 3072
    num_channels: int = 3
    layer_norm_eps: float = 1e-12

    def __post_init__(self):
        if self.hidden_size % self.num_attention_heads:
            raise ValueError(
                f"The hidden size {self.hidden_size} is not a multiple of the number of attention "
                f"heads {self.num_attention_heads}.")
        if self.image_size % self.patch_size:
            raise ValueError("image_size must be a multiple of patch_size")

    # derived shapes (SURVEY.md section 8 notation)
    @property
    def grid(self) -> int:          # g = S / P
        return self.image_size // self.patch_size

    @property
    def num_patches(self) -> int:   # Np = g^2
        return self.grid * self.grid

    @property
    def seq_len(self) -> int:       # N = Np + 1 (CLS)
        return self.num_patches + 1

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def patch_dim(self) -> int:     # K of the patch-embedding GEMM, order (c, py, px)
        return self.num_channels * self.patch_size * self.patch_size

    def forward_flops_per_image(self) -> float:
        """Algorithmic forward FLOPs per image (2*MAC), SURVEY.md section 8(d)."""
        Np, N, D, I, L = self.num_patches, self.seq_len, self.hidden_size, self.intermediate_size, self.num_hidden_layers
        C, S = self.num_classes, self.image_size
        return (2.0 * Np * self.patch_dim * D
                + L * (8.0 * N * D * D + 4.0 * N * N * D + 4.0 * N * D * I)
                + 2.0 * Np * 9 * D * HEAD_MID_CHANNELS + 2.0 * Np * HEAD_MID_CHANNELS * C
                + 8.0 * C * S * S)


# Named shapes used by BASELINE.json's configs (SURVEY.md section 8 "Config sizes").
def vit_tiny16(num_classes=2, image_size=224) -> ViTSegConfig:
    return ViTSegConfig(num_classes, 16, 192, 12, 3, image_size=image_size)


def vit_base16(num_classes=2, image_size=512) -> ViTSegConfig:
    return ViTSegConfig(num_classes, 16, 768, 12, 12, image_size=image_size)


def vit_large16(num_classes=2, image_size=1024) -> ViTSegConfig:
    return ViTSegConfig(num_classes, 16, 1024, 24, 16, image_size=image_size)
