"""Re-run ONE case of tools/fuzz_parity.py (same generator state) and say where its descriptors differ.
python3 tools/fuzz_one.py <seed> <case> [max_w max_h]   (GPU box; POPSIFT_HIP_LIB selects another build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
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
print(fuzz_cases.check_case(O, hip, kw, img))
orc = O.Oracle(O.default_params(**kw), threads=16).run(img)
ctx = hip.Context(hip.default_params(**kw))
ctx.submit(img)
fa, da = orc.fetch()
fb, db = ctx.fetch()
pairs, missing = match_features(fa, fb)
rows = []
for ia, ib in pairs:
    a, b = fa[ia], fb[ib]
    if a["num_ori"] != b["num_ori"]:
        rows.append(("num_ori", int(a["debug_octave"]), float(a["sigma"]), int(a["num_ori"]), int(b["num_ori"])))
        continue
    for k in range(int(a["num_ori"])):
        x, y = da[a["desc_idx"][k]], db[b["desc_idx"][k]]
        rel = float(np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-20))
        if rel > 1e-3:
            rows.append(("desc", int(a["debug_octave"]), round(float(a["xpos"]), 2), round(float(a["ypos"]), 2), round(float(a["sigma"]), 3),
                         k, round(float(a["orientation"][k]), 5), round(float(b["orientation"][k]), 5), round(rel, 4),
                         bool(np.isnan(y).any()), round(float(np.abs(x - y).max()), 4)))
print("missing", missing, "bad rows", len(rows))
for r in rows[:40]:
    print(r)
