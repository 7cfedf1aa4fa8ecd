"""GPU box: one fuzz case (seed, case) in grid mode, descriptors compared in the oracle's frame; lists the offenders.
python3 tools/grid_frame_debug.py <seed> <case> [max_w max_h]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_cases
from oracle import oracle as O
from popsift_amd import _capi as hip
from util import match_features
O.build()
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    kw, img = fuzz_cases.random_case(rng, case, *((int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ()))
print(kw, img.shape, img.dtype)
orc = O.Oracle(O.default_params(**kw), threads=16).run(img)
ctx = hip.Context(hip.default_params(**kw)); ctx.submit(img)
fo, do = orc.fetch(); fh, dh = ctx.fetch()
pairs, _ = match_features(fo, fh)
ori = np.ascontiguousarray(fo["orientation"], np.float32).copy()
for ia, ib in pairs:
    if fo[ia]["num_ori"] == fh[ib]["num_ori"]: ori[ia] = fh[ib]["orientation"]
orc.redo_descriptors(ori)
fo, do = orc.fetch()
up = kw.get("upscale_factor", 1.0)
for ia, ib in pairs:
    a, b = fo[ia], fh[ib]
    if a["num_ori"] != b["num_ori"]: continue
    for k in range(int(a["num_ori"])):
        x, y = do[a["desc_idx"][k]], dh[b["desc_idx"][k]]
        rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-20))
        if rel > 1e-3:
            o = int(a["debug_octave"]); w, h = orc.octave_dims(o)
            sc = 2.0 ** (o - int(up))
            d = (x - y).reshape(16, 8)
            cells = [int(c) for c in np.nonzero(np.abs(d).max(1) > 1e-4)[0]]
            print("octave %d (%dx%d) x %.3f y %.3f sigma %.3f (in octave: %.3f %.3f %.3f) ori[%d] %.7f hip %.7f rel %.2e cells %s" % (
                o, w, h, a["xpos"], a["ypos"], a["sigma"], a["xpos"] / sc, a["ypos"] / sc, a["sigma"] / sc, k,
                a["orientation"][k], b["orientation"][k], rel, cells))
