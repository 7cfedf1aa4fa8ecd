"""Generates tests/golden/*.npz with the CPU oracle (regression fixtures of the oracle itself;
the reference ships no golden vectors and cannot be built here -- parity unpinned).
Run from the repo root:  python tools/make_golden.py
Each fixture holds the input image, the parameters, the sorted features, their descriptors
and a few pyramid planes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from popsift_amd.synth import synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import sorted_features  # noqa: E402

CASES = {
    "default_96x72": (dict(), synth(21, 96, 72)),
    "vlfeat_3oct_112x84": (dict(octaves=3, sift_mode=2), synth(22, 112, 84)),
    "opencv_classic_97x75": (dict(sift_mode=1, gauss_mode=3, norm_mode=1, norm_multi=9), synth(23, 97, 75)),
    "noupscale_levels4_150x110": (dict(upscale_factor=0.0, levels=4), synth(24, 150, 110)),
    "grid_desc_90x70": (dict(desc_mode=2), synth(25, 90, 70)),
    "notile_desc_90x70": (dict(desc_mode=4), synth(26, 90, 70)),
}

if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for name, (kw, img) in CASES.items():
        o = O.Oracle(O.default_params(**kw), threads=4).run(img)
        feats, desc = o.fetch()
        f, d = sorted_features(feats, desc)
        planes = {"g_o0_l%d" % l: o.plane(0, 0, l)[::4, ::4] for l in (0, 3)}
        planes["dog_o1_l2"] = o.plane(1, 1, 2)
        np.savez_compressed(os.path.join(out, name + ".npz"), image=img,
                            params=np.array(sorted(kw.items()), dtype=object).astype(str) if kw else np.zeros((0, 2), str),
                            xpos=f["xpos"], ypos=f["ypos"], sigma=f["sigma"], octave=f["debug_octave"],
                            num_ori=f["num_ori"], orientation=f["orientation"],
                            desc=d, ext_counts=np.array(o.ext_counts()), **planes)
        print(name, len(f), len(d), os.path.getsize(os.path.join(out, name + ".npz")) // 1024, "KiB")
