"""The oracle must reproduce the committed golden fixtures (tests/golden/*.npz, made by
tools/make_golden.py with the oracle itself: they guard the oracle against regressions;
the reference has no golden vectors of its own -- parity unpinned)."""
import glob
import os

import numpy as np
import pytest

from util import sorted_features

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_case(path):
    z = np.load(path, allow_pickle=False)
    kw = {}
    for k, v in z["params"]:
        kw[str(k)] = float(v) if str(k) in ("upscale_factor",) else int(float(v))
    return z, kw


def check_against_golden(z, feats, desc, rtol_pos, rtol_desc):
    f, d = sorted_features(feats, desc)
    assert len(f) == len(z["xpos"]) and len(d) == len(z["desc"])
    assert np.array_equal(f["debug_octave"], z["octave"])
    assert np.array_equal(f["num_ori"], z["num_ori"])
    np.testing.assert_allclose(f["xpos"], z["xpos"], rtol=rtol_pos, atol=1e-5)
    np.testing.assert_allclose(f["ypos"], z["ypos"], rtol=rtol_pos, atol=1e-5)
    np.testing.assert_allclose(f["sigma"], z["sigma"], rtol=1e-5)
    np.testing.assert_allclose(f["orientation"], z["orientation"], atol=2e-3)
    rel = np.linalg.norm(d - z["desc"], axis=1) / np.linalg.norm(z["desc"], axis=1)
    assert np.quantile(rel, 0.995) < rtol_desc and rel.max() < 30 * rtol_desc


def test_fixtures_exist():
    assert len(GOLDEN) >= 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle_mod, path):
    O = oracle_mod
    z, kw = load_case(path)
    o = O.Oracle(O.default_params(**kw), threads=2).run(z["image"])
    assert o.ext_counts() == z["ext_counts"].tolist()
    feats, desc = o.fetch()
    check_against_golden(z, feats, desc, 1e-6, 1e-5)
    for l in (0, 3):
        np.testing.assert_allclose(o.plane(0, 0, l)[::4, ::4], z["g_o0_l%d" % l], rtol=0, atol=1e-4)
    np.testing.assert_allclose(o.plane(1, 1, 2), z["dog_o1_l2"], rtol=0, atol=1e-4)
