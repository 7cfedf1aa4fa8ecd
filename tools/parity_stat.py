"""Share of descriptors within 1e-3 of the oracle (default loop descriptor) over a set of images / modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
from oracle import oracle as O
from util import compare_features
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tot = bad = nori = angbad = unexpl = 0
mx = 0.0
for seed, (w, h) in enumerate([(640, 480), (1920, 1080), (333, 257), (800, 600), (1280, 720)]):
    for kw in (dict(desc_mode=mode), dict(desc_mode=mode, sift_mode=2, norm_mode=1), dict(desc_mode=mode, sift_mode=1, gauss_mode=3)):
        img = synth(400 + seed, w, h)
        orc = O.Oracle(O.default_params(**kw), threads=16).run(img)
        ctx = hip.Context(hip.default_params(**kw)); ctx.submit(img)
        st = compare_features(*orc.fetch(), *ctx.fetch())
        tot += st["n_desc"]; bad += st["desc_bad"]; mx = max(mx, st["max_desc"]); nori += st["num_ori_diff"]; angbad += st["ang_bad"]; unexpl += st["unexplained"]
        ctx.close()
print("desc_mode %d: %d descriptors, %d beyond 1e-3 (%.4f %%), max %.2e; features with a different orientation count %d; "
      "orientations beyond 1e-3 rad %d; descriptors beyond 1e-3 whose orientation AGREES with the oracle's (tests/util.py) %d"
      % (mode, tot, bad, 100.0 * bad / tot, mx, nori, angbad, unexpl))
