"""Known-answer tests that pin the oracle's tables and geometry.
The reference ships no golden vectors (parity unpinned); these KATs are derived from the
formulas in gauss_filter.cu:163-372, popsift.cpp:89-120, sift_conf.cu:275-278."""
import numpy as np
import pytest


def test_default_spans_and_sigmas(oracle_mod):
    O = oracle_mod
    f, span, sig = O.Oracle().gauss_table()
    assert span.tolist() == [6, 6, 8, 9, 11, 14]
    np.testing.assert_allclose(sig, [1.2490, 1.2263, 1.5450, 1.9466, 2.4525, 3.0900], atol=1e-4)
    # inc.sigma[l]^2 = sigma_l^2 - sigma_{l-1}^2 with sigma_l = 1.6 * 2^(l/3)
    for l in range(1, 6):
        want = np.sqrt((1.6 * 2 ** (l / 3)) ** 2 - (1.6 * 2 ** ((l - 1) / 3)) ** 2)
        assert abs(sig[l] - want) < 1e-5
    # level 0: sqrt(1.6^2 - (0.5*2)^2)
    assert abs(sig[0] - np.sqrt(1.6 ** 2 - 1.0)) < 1e-6


def test_filters_normalised_and_truncated(oracle_mod):
    f, span, sig = oracle_mod.Oracle().gauss_table()
    for l in range(6):
        s = float(f[l, 0]) + 2.0 * float(f[l, 1:].astype(np.float64).sum())
        assert abs(s - 1.0) < 1e-6
        assert np.all(f[l, span[l]:] == 0.0)          # taps >= span are zero (kernels read filter[span])
        assert np.all(f[l, :span[l]] > 0.0)
        assert np.all(np.diff(f[l, :span[l]]) < 0.0)   # monotone half kernel
        k = np.arange(span[l])
        g = np.exp(-0.5 * (k / sig[l]) ** 2)
        g /= g[0] + 2 * g[1:].sum()
        np.testing.assert_allclose(f[l, :span[l]], g, rtol=2e-6)


def test_opencv_spans(oracle_mod):
    O = oracle_mod
    _, span, _ = O.Oracle(O.default_params(gauss_mode=3)).gauss_table()
    assert span.tolist() == [6, 6, 7, 9, 11, 14]


@pytest.mark.parametrize("w,h,octaves,bw,bh", [
    (1920, 1080, 9, 3840, 2160),
    (640, 480, 7, 1280, 960),
    (3840, 2160, 10, 7680, 4320),
    (850, 680, 8, 1700, 1360),
    (800, 640, 8, 1600, 1280),
    (17, 13, 2, 34, 26),
])
def test_octave_plan(oracle_mod, w, h, octaves, bw, bh):
    assert oracle_mod.Oracle().plan(w, h) == (octaves, bw, bh)


def test_octave_dims_halve_with_ceil(oracle_mod):
    O = oracle_mod
    o = O.Oracle()
    o.run(np.zeros((1080 // 8, 1920 // 8), np.uint8), keypoints=False)  # 240x135 input
    dims = [o.octave_dims(i) for i in range(o.num_octaves)]
    assert dims[0] == (480, 270)
    for a, b in zip(dims, dims[1:]):
        assert b == ((a[0] + 1) // 2, (a[1] + 1) // 2)


def test_full_1080p_pyramid_pixel_count(oracle_mod):
    # sum over octaves of w*h = 11 059 245 for the 1080p default config
    w, h, tot = 3840, 2160, 0
    for _ in range(9):
        tot += w * h
        w, h = (w + 1) // 2, (h + 1) // 2
    assert tot == 11059245
    assert (w, h) == (8, 5)  # octave 8 is 15x9, the next would be 8x5


def test_forced_octaves_and_downsampling(oracle_mod):
    O = oracle_mod
    assert O.Oracle(O.default_params(octaves=3)).plan(640, 480) == (3, 1280, 960)
    assert O.Oracle(O.default_params(upscale_factor=0.0)).plan(640, 480) == (6, 640, 480)
    assert O.Oracle(O.default_params(upscale_factor=-1.0)).plan(640, 480)[1:] == (320, 240)


def test_rejects_unsupported(oracle_mod):
    O = oracle_mod
    for kw in (dict(sigma=2.5), dict(levels=10), dict(gauss_mode=1), dict(desc_mode=5), dict(desc_mode=-1)):
        with pytest.raises(ValueError):
            O.Oracle(O.default_params(**kw))


def test_solve3_matches_linear_algebra(oracle_mod):
    rng = np.random.default_rng(0)
    for _ in range(50):
        m = rng.normal(size=(3, 3)).astype(np.float32)
        A = (m + m.T).astype(np.float32)          # symmetric like the DoG Hessian
        b = rng.normal(size=3).astype(np.float32)
        ok, x = oracle_mod.solve3(A, b)
        assert ok
        want = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        np.testing.assert_allclose(x, want, rtol=2e-3, atol=2e-4)
    ok, _ = oracle_mod.solve3(np.zeros((3, 3), np.float32), np.ones(3, np.float32))
    assert not ok   # det == 0 -> d = 0, s_solve.h:53-55


def test_normalize_modes(oracle_mod):
    rng = np.random.default_rng(1)
    v = rng.random(128).astype(np.float32) * 50
    rs = oracle_mod.normalize(v, 0, 0)
    np.testing.assert_allclose(rs, np.sqrt(v / v.sum()), rtol=1e-5)
    assert abs(float((rs.astype(np.float64) ** 2).sum()) - 1.0) < 1e-5
    l2 = oracle_mod.normalize(v, 1, 0)
    c = np.minimum(v, 0.2 * np.linalg.norm(v))
    np.testing.assert_allclose(l2, c / np.linalg.norm(c), rtol=1e-5)
    assert abs(np.linalg.norm(l2) - 1.0) < 1e-5
    # norm_multi multiplies by 2^m
    np.testing.assert_allclose(oracle_mod.normalize(v, 0, 9), rs * 512.0, rtol=1e-6)
    np.testing.assert_allclose(oracle_mod.normalize(v, 1, 8), l2 * 256.0, rtol=1e-6)
