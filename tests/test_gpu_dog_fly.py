"""params.store_dog = 0 (the default: DoG planes not stored; detection / refinement subtract the Gaussian planes
they load) must give bit-identical planes, extrema, features and descriptors to the stored-DoG path
(store_dog = 1, what the reference does).  Each mode runs in its own child process on the same seeded images."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
out = {}
for tag, spec, kw in (("a", (5, 640, 480), {}), ("b", (9, 333, 251), {"sift_mode": 1, "octaves": 3}),
                      ("c", (11, 400, 300), {"sift_mode": 2})):
    ctx = hip.Context(hip.default_params(store_dog=int(sys.argv[3]), **kw))
    ctx.submit(synth(*spec))
    feats, desc = ctx.fetch()
    # canonical order (the device's compaction order is arbitrary): octave, y, x, sigma
    order = np.lexsort((feats["sigma"], feats["xpos"], feats["ypos"], feats["debug_octave"]))
    f = feats[order]
    out[tag + "_pos"] = np.stack([f["xpos"], f["ypos"], f["sigma"]], 1)
    out[tag + "_nori"] = f["num_ori"]
    out[tag + "_ori"] = np.concatenate([r["orientation"][: int(r["num_ori"])] for r in f]) if len(f) else np.zeros(0, np.float32)
    out[tag + "_desc"] = np.array([desc[r["desc_idx"][k]] for r in f for k in range(int(r["num_ori"]))], np.float32).reshape(-1, 128)
    e = ctx.extrema()
    eo = np.lexsort((e["xpos"], e["ypos"], e["lpos"], e["octave"]))
    out[tag + "_ext"] = np.stack([e["octave"][eo].astype(np.float32), e["lpos"][eo].astype(np.float32), e["xpos"][eo], e["ypos"][eo]], 1)
    out[tag + "_dog"] = ctx.plane(0, 1, 2)
    out[tag + "_dog1"] = ctx.plane(1, 1, 0)
np.savez(sys.argv[2], **out)
"""


def _run(mode, path):
    subprocess.run([sys.executable, "-c", CHILD, ROOT, path, str(mode)], check=True, timeout=600)
    return np.load(path)


@pytest.mark.gpu
def test_dog_on_the_fly_is_bit_identical(tmp_path):
    stored = _run(1, str(tmp_path / "stored.npz"))
    fly = _run(0, str(tmp_path / "fly.npz"))
    assert sorted(stored.files) == sorted(fly.files)
    for k in stored.files:
        assert stored[k].shape == fly[k].shape, k
        assert np.array_equal(stored[k].view(np.uint32) if stored[k].dtype == np.float32 else stored[k],
                              fly[k].view(np.uint32) if fly[k].dtype == np.float32 else fly[k]), k
    assert stored["a_desc"].shape[0] > 1000
