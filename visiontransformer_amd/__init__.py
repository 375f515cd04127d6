"""MI355X-native ViT-segmentation hot path of mtumalan/VisionTransformer."""
from .config import ViTSegConfig, vit_base16, vit_large16, vit_tiny16  # noqa: F401
from .model import ViTSegmentationModel  # noqa: F401
from .lightning import LightningViTModel  # noqa: F401
from .predict import predict, load_model  # noqa: F401
