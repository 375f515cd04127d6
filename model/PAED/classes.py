"""Entry-point shim for the reference's model/PAED scripts (`from classes import ...`, e.g.
model/PAED/ViTscript.py:8,66): the names resolve to the MI355X implementation."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visiontransformer_amd import ViTSegmentationModel  # noqa: E402,F401
from visiontransformer_amd.paed import LightningViTModel, PAEDTrainer, paed_loss_multiclass_soft  # noqa: E402,F401
