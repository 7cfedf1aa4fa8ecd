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
# optional Params overrides: PROF_KW='{"filter_max_extrema": 20000, "filter_sorting": 1, "filter_grid_size": 4}'
import json
kw = json.loads(os.environ.get("PROF_KW", "{}"))
ctx = hip.Context(hip.default_params(**kw))
for item in filter(None, os.environ.get("PROF_DEBUG", "").split(",")):   # PROF_DEBUG="6:0,5:16": popsift_hip_debug_set(what, value)
    what, value = item.split(":")
    ctx.debug_set(int(what), int(value))
for i in range(n):
    ctx.submit(img); c = ctx.wait(); print(c, "%.3f ms" % ctx.report().ms_device, flush=True)
