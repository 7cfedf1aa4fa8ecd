"""BASELINE.json's full sizes against the oracle: config 2 (1920x1080) and config 3 (3840x2160, base plane 7680x4320,
10 octaves: the largest single-GPU configuration) -- sampled planes bit for bit, the full extrema set, features and
descriptors under the unified bars of util.feature_parity -- plus config 3's size-independent properties."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from util import bits, feature_parity, sorted_features

pytestmark = pytest.mark.gpu


def test_config2_1080p_against_oracle(oracle_mod, gpu_hip):
    img = synth(2, 1920, 1080)
    orc = oracle_mod.Oracle(threads=16).run(img)
    ctx = gpu_hip.Context()
    ctx.submit(img)
    rep_counts = list(ctx.report().ext_ct)
    nf, nd = ctx.wait()
    assert ctx.report().num_octaves == 9 and (ctx.report().base_w, ctx.report().base_h) == (3840, 2160)
    assert abs(ctx.report().pyramid_pixels - 11059245) < 0.5
    assert (nf, nd) == orc.counts()
    for o, l in ((0, 5), (1, 3), (4, 2), (8, 5)):
        assert np.array_equal(bits(orc.plane(o, 0, l)), bits(ctx.plane(o, 0, l)))
    for o, l in ((0, 4), (2, 0), (8, 4)):
        assert np.array_equal(bits(orc.plane(o, 1, l)), bits(ctx.plane(o, 1, l)))
    eo, eh = orc.extrema(), ctx.extrema()
    key = lambda e: sorted(zip(e["octave"].tolist(), e["lpos"].tolist(), e["xpos"].tolist(), e["ypos"].tolist()))
    assert key(eo) == key(eh)                       # the full extrema set, bit-exact refined positions
    ok, msg, st = feature_parity(orc, *ctx.fetch())   # the unified bars (util.feature_parity), orientations included
    assert ok, msg


def test_config3_4k_against_oracle(oracle_mod, gpu_hip):
    img = synth(3, 3840, 2160)
    # the default cap of 100000 extrema per octave is hit by octave 0 of this image; WHICH extrema survive the cap is
    # arrival order (as in the reference, s_extrema.cu:541), so lift the cap on both sides
    kw = dict(max_extrema=400000)
    orc = oracle_mod.Oracle(oracle_mod.default_params(**kw), threads=16).run(img)
    ctx = gpu_hip.Context(gpu_hip.default_params(**kw))
    ctx.submit(img)
    nf, nd = ctx.wait()
    rep = ctx.report()
    assert rep.num_octaves == orc.num_octaves == 10 and (rep.base_w, rep.base_h) == (7680, 4320)
    assert max(rep.ext_ct) < 400000 and (nf, nd) == orc.counts()
    for o, l in ((0, 0), (0, 5), (1, 3), (2, 1), (5, 2), (9, 5)):
        assert np.array_equal(bits(orc.plane(o, 0, l)), bits(ctx.plane(o, 0, l))), (o, l)
    for o, l in ((0, 2), (3, 0), (9, 4)):
        assert np.array_equal(bits(orc.plane(o, 1, l)), bits(ctx.plane(o, 1, l))), (o, l)
    eo, eh = orc.extrema(), ctx.extrema()
    key = lambda e: sorted(zip(e["octave"].tolist(), e["lpos"].tolist(), e["xpos"].tolist(), e["ypos"].tolist()))
    assert key(eo) == key(eh)
    ok, msg, st = feature_parity(orc, *ctx.fetch())
    assert ok, msg
    assert st["n_a"] > 250000


def test_config3_4k_properties(gpu_hip):
    img = synth(3, 3840, 2160)
    # the default cap of 100000 extrema per octave is hit by octave 0 of this image; WHICH extrema
    # survive the cap is arrival order (as in the reference, s_extrema.cu:541), so lift the cap
    ctx = gpu_hip.Context(gpu_hip.default_params(max_extrema=400000))
    feats, desc = ctx.submit(img).fetch()
    assert max(ctx.report().ext_ct) < 400000
    rep = ctx.report()
    assert rep.num_octaves == 10 and (rep.base_w, rep.base_h) == (7680, 4320)
    assert abs(rep.pyramid_pixels - 44236845) < 0.5
    assert len(feats) == rep.ext_total > 10000 and len(desc) == rep.ori_total == int(feats["num_ori"].sum())
    # RootSift: sum of squares = 1; all components >= 0
    assert desc.min() >= 0.0
    np.testing.assert_allclose((desc.astype(np.float64) ** 2).sum(1), 1.0, rtol=2e-4)
    assert feats["xpos"].min() >= 0 and feats["xpos"].max() <= 3840 and feats["ypos"].max() <= 2160
    assert feats["num_ori"].min() >= 1 and feats["num_ori"].max() <= 4
    # DoG = difference of neighbouring Gaussian planes, bit for bit, on the largest octave
    g4, g5, d4 = ctx.plane(0, 0, 4), ctx.plane(0, 0, 5), ctx.plane(0, 1, 4)
    assert np.array_equal(bits(g5 - g4), bits(d4))
    # octave 1 level 0 is octave 0 level 3 sampled at every second pixel
    assert np.array_equal(bits(ctx.plane(1, 0, 0)), bits(ctx.plane(0, 0, 3)[::2, ::2]))
    # idempotence: a second run on the same context reproduces the result bit for bit
    f2, d2 = sorted_features(*ctx.submit(img).fetch())
    f1, d1 = sorted_features(feats, desc)
    assert np.array_equal(f1["xpos"], f2["xpos"]) and np.array_equal(bits(d1), bits(d2))
    # cropping invariance: keypoints far from the cut are unchanged when the right half is removed
    left = np.ascontiguousarray(img[:, :1920])
    fl, _ = gpu_hip.Context(gpu_hip.default_params(max_extrema=400000)).submit(left).fetch()
    sel = (feats["debug_octave"] == 0) & (feats["xpos"] < 1700)
    a = set(zip(feats["xpos"][sel].tolist(), feats["ypos"][sel].tolist()))
    sel_l = (fl["debug_octave"] == 0) & (fl["xpos"] < 1700)
    b = set(zip(fl["xpos"][sel_l].tolist(), fl["ypos"][sel_l].tolist()))
    assert len(a ^ b) <= max(2, len(a) // 2000)
