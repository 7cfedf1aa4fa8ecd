"""Every way the library can build a Gaussian plane gives the oracle's plane, bit for bit.

The pyramid's plane-to-plane levels have two sets of kernels: one tile per workgroup (pyramid.hip) and the strip march
(blur_march.hip) that the large planes take.  Which one a launch uses depends on the plane's size and on the batch -- a
matter of speed only; the smallest octaves are built by one launch of one workgroup per image (pyr_tail.hip) or, with
PYR_TAIL = 1, by level launches like the others.  The debug switch BLUR_PATH forces either (1 = tiles only, 2 = march wherever it applies), BLUR_SEG
the rows per segment of the march, so the small images the oracle finishes in seconds reach the march kernels too:
strips cut by the plane's border, segments of one and several steps, a last step of fewer than 32 rows, planes of fewer
rows than the filter is long, every HALO instance (spans of levels 2 .. 6 at sigma 1 .. 2, opencv spans)."""
import numpy as np
import pytest

from popsift_amd.synth import synth
from util import bits, sorted_features

pytestmark = pytest.mark.gpu

BLUR_PATH, BLUR_SEG, PYR_TAIL = 8, 9, 10

CASES = [
    ("default_400x300", dict(), (61, 400, 300)),
    ("width_not_multiple_of_128_333x257", dict(), (62, 333, 257)),
    ("one_strip_60x200", dict(), (63, 60, 200)),
    ("tall_9x300", dict(), (42, 9, 300)),
    ("thin_300x9", dict(), (25, 300, 9)),
    ("tiny_17x13", dict(), (20, 17, 13)),
    ("no_upscale_257x129", dict(upscale_factor=0.0), (41, 257, 129)),
    ("levels5_sigma1p3", dict(levels=5, sigma=1.3), (64, 260, 200)),
    ("levels2_sigma2", dict(levels=2, sigma=2.0), (65, 220, 170)),
    ("levels6_sigma1", dict(levels=6, sigma=1.0), (66, 200, 260)),
    ("opencv_gauss", dict(sift_mode=1, gauss_mode=3), (67, 300, 220)),
    ("float_input", dict(), (68, 280, 210)),
    ("deep_octaves_40x24", dict(octaves=7), (40, 40, 24)),            # planes shrink to 2 x 1: all but octave 0 in the tail
    ("tail_from_a_plane_that_just_fits_256x176", dict(upscale_factor=0.0), (69, 256, 176)),  # octave 1 = 128 x 88
]


def planes(ctx, n_oct, L):
    return [[ctx.plane(o, 0, l) for l in range(L)] for o in range(n_oct)]


@pytest.mark.parametrize("name,kw,spec", CASES, ids=[c[0] for c in CASES])
def test_march_and_tile_kernels_give_the_oracles_planes(oracle_mod, gpu_hip, name, kw, spec):
    img = synth(*spec)
    if name == "float_input":
        img = img.astype(np.float32) / 256.0
    L = max(2, kw.get("levels", 3)) + 3
    orc = oracle_mod.Oracle(oracle_mod.default_params(**kw), threads=8).run(img)
    want = [[orc.plane(o, 0, l) for l in range(L)] for o in range(orc.num_octaves)]
    # (blur path, rows per march segment, PYR_TAIL: 1 = no tail launch)
    for path, seg, tail in ((1, 0, 1), (1, 0, 0), (2, 0, 0), (2, 32, 1), (2, 64, 0), (2, 96, 0)):
        ctx = gpu_hip.Context(gpu_hip.default_params(**kw))
        ctx.debug_set(BLUR_PATH, path)
        ctx.debug_set(BLUR_SEG, seg)
        ctx.debug_set(PYR_TAIL, tail)
        ctx.submit(img)
        ctx.wait()
        assert ctx.report().num_octaves == orc.num_octaves
        got = planes(ctx, orc.num_octaves, L)
        for o in range(orc.num_octaves):
            for l in range(L):
                a, b = want[o][l], got[o][l]
                assert np.array_equal(bits(a), bits(b)), "path %d seg %d tail %d octave %d level %d: %d values differ, max %g" % (
                    path, seg, tail, o, l, int((bits(a) != bits(b)).sum()), float(np.abs(a - b).max()))
        ctx.close()


def _canon(feats, desc):
    f, d = sorted_features(feats, desc)
    return (bits(f["xpos"]), bits(f["ypos"]), bits(f["sigma"]), f["num_ori"].copy(), bits(f["orientation"]), bits(d))


def test_march_in_a_batch(gpu_hip):
    """blockIdx.y = image: every image of a batch through the march kernels equals its own single submit through the
    tile kernels (features, orientations and descriptors bit for bit)."""
    imgs = [synth(70 + k, 384, 288) for k in range(3)]
    want = []
    for im in imgs:
        ctx = gpu_hip.Context()
        ctx.debug_set(BLUR_PATH, 1)
        want.append(_canon(*ctx.submit(im).fetch()))
        ctx.close()
    ctx = gpu_hip.Context()
    ctx.debug_set(BLUR_PATH, 2)
    ctx.debug_set(BLUR_SEG, 64)
    ctx.submit_batch(imgs)
    ctx.wait_batch()
    for k in range(3):
        got = _canon(*ctx.fetch_item(k))
        assert all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(got, want[k])), "image %d" % k
    ctx.close()


def test_the_batch_path_of_the_timed_loop_at_full_size(gpu_hip):
    """Four 1080p images in one submit: 4 x 8.3 Mpx per level launch of octave 0, so launch_blur takes the march kernels by
    itself (no debug switch: the path of bench.py's timed loop, equal segments of 224 rows) -- every image equals its own single
    submit, which the tile kernels build (features, orientations, descriptors and two planes bit for bit)."""
    imgs = [synth(100 + k, 1920, 1080) for k in range(4)]
    want, planes = [], []
    single = gpu_hip.Context()
    for im in imgs:
        want.append(_canon(*single.submit(im).fetch()))
        planes.append((single.plane(0, 0, 5).copy(), single.plane(1, 0, 2).copy()))
    single.close()
    ctx = gpu_hip.Context()
    ctx.submit_batch(imgs)
    assert len(ctx.wait_batch()) == 4
    for k in range(4):
        got = _canon(*ctx.fetch_item(k))
        assert all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(got, want[k])), "image %d" % k
    # the planes of image 0 of the batch (the debug download reads slot 0)
    assert np.array_equal(bits(ctx.plane(0, 0, 5)), bits(planes[0][0])) and np.array_equal(bits(ctx.plane(1, 0, 2)), bits(planes[0][1]))
    ctx.close()
