"""Device time (first to last kernel, HIP events) of one image for the BASELINE.json configurations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth

CONFIGS = [
    ("config 1: 640x480, 3 octaves, VLFeat mode", dict(octaves=3, sift_mode=2), (1, 640, 480)),
    ("config 2: 1920x1080 default", dict(), (2, 1920, 1080)),
    ("config 3: 3840x2160 default (base 7680x4320)", dict(max_extrema=400000), (3, 3840, 2160)),
    ("config 5 stand-in: 850x680 synthetic, OpenCV mode + opencv Gauss", dict(sift_mode=1, gauss_mode=3), (200, 850, 680)),
    ("config 2 with grid descriptor", dict(desc_mode=2), (2, 1920, 1080)),
    ("config 2 with notile descriptor", dict(desc_mode=4), (2, 1920, 1080)),
    ("config 2 with grid filter 20000 / 4x4 / largest first", dict(filter_max_extrema=20000, filter_sorting=1, filter_grid_size=4), (2, 1920, 1080)),
]
for name, kw, spec in CONFIGS:
    img = synth(*spec)
    ctx = hip.Context(hip.default_params(**kw))
    ms = []
    for _ in range(6):
        ctx.submit(img); nf, nd = ctx.wait(); ms.append(ctx.report().ms_device)
    r = ctx.report()
    b_alg = img.size + 4.0 * r.pyramid_pixels * 22 + 52.0 * nf + 512.0 * nd
    t = float(np.median(ms[1:]))
    print("%-66s T_dev %7.3f ms  %8.1f Mpix/s  features %7d descriptors %7d  B_alg/T %6.0f GB/s" % (
        name, t, img.size / 1e6 / (t * 1e-3), nf, nd, b_alg / (t * 1e-3) / 1e9), flush=True)
    ctx.close()
