"""GPU box: one 3840x2160 image (config 3, base plane 7680x4320) with the tile kernels and with the march kernels forced
(debug switch 8 = 2) at several segment heights (switch 9): T_dev by HIP events.  python3 tools/r04_config3_march.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth

img = synth(3, 3840, 2160)
for name, dbg in (("tile kernels (default for one image)", ()), ("march, library's segments", ((8, 2),)),
                  ("march, 160 rows", ((8, 2), (9, 160))), ("march, 224 rows", ((8, 2), (9, 224))), ("march, 320 rows", ((8, 2), (9, 320))),
                  ("march, 480 rows", ((8, 2), (9, 480)))):
    ctx = hip.Context(hip.default_params(max_extrema=400000))
    for w, v in dbg:
        ctx.debug_set(w, v)
    ms = []
    for _ in range(6):
        ctx.submit(img); ctx.wait(); ms.append(ctx.report().ms_device)
    print("%-40s T_dev %.3f ms" % (name, float(np.median(ms[1:]))), flush=True)
    ctx.close()
