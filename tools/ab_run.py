#!/usr/bin/env python3
"""Same-box A/B of two builds of libvitseg (boxes of the pool differ by up to 12 %, so before/after pairs must share one
gpurun call): copy the old build to visiontransformer_amd/csrc/libvitseg_prev.so, rebuild, then
    python tools/ab_run.py prev <op_probe arguments>     # runs tools/op_probe.py against the old library
    python tools/ab_run.py new  <op_probe arguments>     # ... against the current one"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visiontransformer_amd._lib as L  # noqa: E402

if sys.argv[1] == "prev":
    L.LIB_PATH = L.LIB_PATH.replace("libvitseg.so", "libvitseg_prev.so")
sys.argv = ["op_probe.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "op_probe.py"), run_name="__main__")
