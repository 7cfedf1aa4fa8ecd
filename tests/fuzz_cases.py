"""Randomised differential cases oracle vs HIP (random sizes / parameters; planes and extrema bit-exact, descriptors
by tolerance).  tests/test_gpu_fuzz.py runs a fixed-seed slice in the GPU suite; tools/fuzz_parity.py runs as many as
asked for."""
import time

import numpy as np

from popsift_amd.synth import synth
from util import bits, feature_parity


def random_case(rng, case, max_w=700, max_h=500):
    w = int(rng.integers(9, max_w))
    h = int(rng.integers(9, max_h))
    kw = dict(levels=int(rng.integers(2, 7)), sigma=float(np.float32(rng.uniform(1.0, 2.0))),
              sift_mode=int(rng.integers(0, 3)), gauss_mode=int(rng.choice([0, 3])),
              upscale_factor=float(rng.choice([1.0, 1.0, 0.0, -1.0])), norm_mode=int(rng.integers(0, 2)),
              norm_multi=int(rng.choice([0, 0, 9])), desc_mode=int(rng.choice([0, 0, 0, 1, 2, 3, 4])),
              edge_limit=float(np.float32(rng.uniform(5, 15))), threshold=float(np.float32(rng.uniform(0.02, 0.08))),
              max_extrema=int(rng.choice([100000, 100000, 300])))
    if rng.random() < 0.3:
        kw["octaves"] = int(rng.integers(1, 6))
    if rng.random() < 0.25 and kw["max_extrema"] != 300:   # a binding cap keeps an arrival-order subset: not comparable
        kw.update(filter_max_extrema=int(rng.integers(50, 600)), filter_sorting=int(rng.integers(1, 3)),
                  filter_grid_size=int(rng.integers(1, 6)))
    img = synth(1000 + case, w, h)
    if rng.random() < 0.3:
        img = (img.astype(np.float32) / 256.0)
    return kw, img


def check_case(O, hip, kw, img, threads=16, debug=()):
    """Returns (ok, message).  debug: popsift_hip_debug_set switches (what, value) set on the context before the submit --
    e.g. ((8, 2), (9, 64)) sends every plane-to-plane level through the march kernels in 64-row segments."""
    t0 = time.time()
    try:
        orc = O.Oracle(O.default_params(**kw), threads=threads).run(img)
        ctx = hip.Context(hip.default_params(**kw))
        for what, value in debug:
            ctx.debug_set(what, value)
        ctx.submit(img)
        ctx.wait()
        ok = ctx.report().num_octaves == orc.num_octaves
        L = max(2, kw["levels"]) + 3
        for o in range(orc.num_octaves):
            for l in range(L):
                ok = ok and np.array_equal(bits(orc.plane(o, 0, l)), bits(ctx.plane(o, 0, l)))
            for l in range(L - 1):
                ok = ok and np.array_equal(bits(orc.plane(o, 1, l)), bits(ctx.plane(o, 1, l)))
        planes_ok = ok
        capped = kw["max_extrema"] == 300 and max(orc.ext_counts() + [0]) >= 300
        msg = ""
        if not capped:  # with a binding max_extrema cap the surviving subset depends on arrival order
            eo, eh = orc.extrema(), ctx.extrema()
            key = lambda e: sorted(zip(e["octave"].tolist(), e["lpos"].tolist(), e["xpos"].tolist(), e["ypos"].tolist()))
            ext_ok = key(eo) == key(eh)
            ok = ok and ext_ok
            if ext_ok:
                # the bars of the named cases (util.feature_parity; grid descriptors in the oracle's frame)
                dok, pmsg, st = feature_parity(orc, *ctx.fetch(), grid_mode=kw["desc_mode"] == 2)
                n = max(st["n_desc"], 1)
                ok = ok and dok
                msg = pmsg if not dok else "desc_bad %d/%d max %.1e" % (st["desc_bad"], n, st["max_desc"])
            else:
                msg = "EXTREMA DIFFER %d vs %d" % (len(eo), len(eh))
        else:
            msg = "capped"
        ctx.close()
    except Exception as e:  # noqa: BLE001 -- a crash in either side is a failed case, with its parameters printed
        ok, planes_ok, msg = False, False, "EXCEPTION %r" % (e,)
    return ok, "planes %s %s (%.1fs)" % (planes_ok, msg, time.time() - t0)
