"""Pyramid stage of the oracle against independent implementations."""
import numpy as np
import pytest
from scipy import ndimage

from popsift_amd.synth import synth


def _taps(f, span, l):
    half = f[l, :span[l]].astype(np.float64)
    return np.concatenate([half[:0:-1], half])


def test_level0_is_bilinear_2x_upscale_then_blur(oracle_mod):
    O = oracle_mod
    img = synth(3, 96, 64)
    o = O.Oracle().run(img, keypoints=False)
    f, span, _ = o.gauss_table()
    # PopSift/VLFeat mode, upscale 1: source x = X/2 exactly -> even pixels copy, odd average
    src = img.astype(np.float64) / 255.0
    up = np.zeros((128, 192))
    up[::2, ::2] = src
    up[::2, 1::2] = 0.5 * (src + np.roll(src, -1, axis=1))
    up[::2, -1] = src[:, -1]
    up[1::2, :] = 0.5 * (up[::2, :] + np.roll(up[::2, :], -1, axis=0))
    up[-1, :] = up[-2, :]
    k = _taps(f, span, 0)
    ref = ndimage.correlate1d(up, k, axis=1, mode="nearest") * 255.0
    ref = ndimage.correlate1d(ref, k, axis=0, mode="nearest")
    np.testing.assert_allclose(o.plane(0, 0, 0), ref, rtol=0, atol=2e-3)


def test_incremental_blur_matches_scipy(oracle_mod):
    O = oracle_mod
    img = synth(4, 80, 60)
    o = O.Oracle().run(img, keypoints=False)
    f, span, _ = o.gauss_table()
    for oc in range(o.num_octaves):
        for l in range(1, 6):
            prev = o.plane(oc, 0, l - 1).astype(np.float64)
            k = _taps(f, span, l)
            ref = ndimage.correlate1d(prev, k, axis=1, mode="nearest")
            ref = ndimage.correlate1d(ref.astype(np.float32).astype(np.float64), k, axis=0, mode="nearest")
            np.testing.assert_allclose(o.plane(oc, 0, l), ref, rtol=0, atol=2e-4)


def test_decimation_and_dog(oracle_mod):
    O = oracle_mod
    img = synth(5, 75, 51)  # odd sizes exercise the min(2x, w-1) clamp
    o = O.Oracle().run(img, keypoints=False)
    for oc in range(1, o.num_octaves):
        prev = o.plane(oc - 1, 0, 3)
        cur = o.plane(oc, 0, 0)
        h, w = cur.shape
        ys = np.minimum(2 * np.arange(h), prev.shape[0] - 1)
        xs = np.minimum(2 * np.arange(w), prev.shape[1] - 1)
        assert np.array_equal(cur, prev[np.ix_(ys, xs)])
    for oc in range(o.num_octaves):
        for l in range(5):
            assert np.array_equal(o.plane(oc, 1, l), o.plane(oc, 0, l + 1) - o.plane(oc, 0, l))


def test_constant_image_stays_constant(oracle_mod):
    O = oracle_mod
    img = np.full((40, 56), 77, np.uint8)
    o = O.Oracle().run(img)
    for oc in range(o.num_octaves):
        for l in range(6):
            np.testing.assert_allclose(o.plane(oc, 0, l), 77.0, atol=2e-4)
    assert o.counts() == (0, 0)


def test_float_input_equals_byte_input_scaled(oracle_mod):
    O = oracle_mod
    img = synth(6, 64, 48)
    a = O.Oracle().run(img, keypoints=False)
    b = O.Oracle().run((img.astype(np.float32) / 255.0), keypoints=False)
    np.testing.assert_allclose(a.plane(0, 0, 0), b.plane(0, 0, 0), atol=1e-3)


@pytest.mark.parametrize("threads", [1, 3])
def test_threads_do_not_change_results(oracle_mod, threads):
    O = oracle_mod
    img = synth(7, 120, 90)
    a = O.Oracle(threads=1).run(img)
    b = O.Oracle(threads=threads).run(img)
    fa, da = a.fetch()
    fb, db = b.fetch()
    assert np.array_equal(fa, fb) and np.array_equal(da, db)
