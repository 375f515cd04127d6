#!/usr/bin/env python3
"""Same-box A/B of two builds of libvitseg (boxes of the pool differ by up to 12 %, so before/after pairs must share one
gpurun call): copy the old build to visiontransformer_amd/csrc/libvitseg_prev.so, rebuild, then
    python tools/ab_run.py prev <op_probe arguments>            # tools/op_probe.py against the old library
    python tools/ab_run.py new  <op_probe arguments>            # ... against the current one
    python tools/ab_run.py prev bench --mode train ...          # `bench` as first argument: bench.py instead"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import visiontransformer_amd._lib as L  # noqa: E402

if sys.argv[1] == "prev":
    L.LIB_PATH = L.LIB_PATH.replace("libvitseg.so", "libvitseg_prev.so")
    import ctypes
    _old = ctypes.CDLL(L.LIB_PATH)          # an older build may lack the newest entry points: check only what it has
    L.EXPORTS = [n for n in L.EXPORTS if hasattr(_old, n)]

    class _Missing:                         # stands in for an entry point the old build does not have (argtypes are set on it)
        argtypes = restype = None

    class _Lenient(ctypes.CDLL):
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                if name.startswith("vitseg_"):
                    return _Missing()
                raise

    ctypes.CDLL = _Lenient
rest = sys.argv[2:]
if rest and rest[0] == "bench":
    script, rest = os.path.join(ROOT, "bench.py"), rest[1:]
else:
    script = os.path.join(ROOT, "tools", "op_probe.py")
sys.argv = [os.path.basename(script)] + rest
runpy.run_path(script, run_name="__main__")
