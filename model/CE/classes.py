"""Entry-point shim: the reference's scripts do `from classes import ...` (model/CE/*.py); these
names resolve to the MI355X implementation.  Dataset / smp classes of the reference's classes.py are
host-side I/O and are out of scope."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visiontransformer_amd import LightningViTModel, ViTSegmentationModel  # noqa: E402,F401
