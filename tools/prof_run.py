"""Profiling driver: N extractions of the config-2 synthetic image on one context."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
img = synth(2, W, H)
ctx = hip.Context()
for i in range(n):
    ctx.submit(img); c = ctx.wait(); print(c, "%.3f ms" % ctx.report().ms_device, flush=True)
