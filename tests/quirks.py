"""Crafted scale-space planes that isolate behaviours of the reference a natural image hardly ever decides
(SURVEY.md Appendix A.9).  Used by the oracle KATs (tests/test_oracle_quirks.py) and, uploaded through the C-ABI
debug hook, by the GPU parity test of the same cases (tests/test_gpu_quirks.py).

One octave of W x H (upscale 0, octaves = 1).  The DoG planes hold a paraboloid
    D(x, y, z) = A - a ((x - xc)^2 + (y - yc)^2) - c (z - zc)^2        (floored far away by a plateau)
whose finite differences are exact, so the refinement lands on (xc, yc, zc) up to float rounding; the Gaussian
planes (orientation / descriptor input) are set independently of it.
"""
import numpy as np

W, H = 64, 48
PLATEAU = -1000.0


def dog_planes(xc, yc, zc, A, a, c, n=5):
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = []
    for z in range(n):
        d = A - a * ((xx - xc) ** 2 + (yy - yc) ** 2) - c * (z - zc) ** 2
        out.append(np.maximum(d, PLATEAU).astype(np.float32))
    return out


def gauss_planes(xc, yc, flat_radius=11.0, bump=None, n=6):
    """Constant inside `flat_radius` of the keypoint (no gradient in the orientation window), a radial ramp outside
    (so that the descriptor, whose patch is larger, is not all zero); bump = (x, y, value) adds to one pixel."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    r = np.hypot(xx - xc, yy - yc)
    g = (100.0 + 2.0 * np.maximum(r - flat_radius, 0.0)).astype(np.float32)
    if bump is not None:
        g[bump[1], bump[0]] += np.float32(bump[2])
    return [g.copy() for _ in range(n)]


def params_kw(**kw):
    d = dict(upscale_factor=0.0, octaves=1)
    d.update(kw)
    return d


# case 1 (s_orientation.cu:183-231): an extremum whose orientation window holds no gradient at all has no histogram
#   peak; every yval is -inf, "best >= 0.8 * best" holds for all four sorted entries, and the keypoint gets FOUR
#   orientations, each from refined bin -1: angle = 2 pi (-1) / 36 - pi.
NO_PEAK = dict(dog=dict(xc=30.0, yc=24.0, zc=1.0, A=50.0, a=1.0, c=4.0), gauss=dict(xc=30.0, yc=24.0))
NO_PEAK_ANGLE = float(np.float32(np.float32(2.0 * np.pi) * np.float32(-1.0)) * np.float32(1.0 / 36.0) - np.float32(np.pi))

# case 2 (s_orientation.cu:123): `int sq_dist = dx*dx + dy*dy` is TRUNCATED before the comparison with rad^2.
#   sigma = 1.6 * 2^(1/3) -> rad = round(4.5 sigma) = 9.  The keypoint sits at x = 30.03: the pixel at x = 21 has
#   dx^2 = 81.54 -> 81 <= 81 is inside the window (a float comparison would leave it out).  It is the only pixel
#   of the window with a gradient (the bump at x = 20 gives dx = I(22) - I(20) = -10 there), so the keypoint has
#   exactly ONE orientation, pointing along -x (angle -pi); without the truncation it would be case 1.
TRUNCATION = dict(dog=dict(xc=30.03, yc=24.0, zc=1.0, A=50.0, a=1.0, c=4.0),
                  gauss=dict(xc=30.03, yc=24.0, bump=(20, 24, 10.0)))

# case 3 (s_extrema.cu:150-153): in OpenCV mode the FIRST contrast test compares with floorf(threshold) = 1.0
#   (threshold = 0.04 * 0.5 * 255 / 3 = 1.7).  The candidate pixel holds 1.58 -- below 1.7, above 1.0 -- while the
#   interpolated extremum 0.4 px away reaches 3.5 >= 2 * 1.7: OpenCV mode keeps it; PopSift mode (first test
#   1.6 * 1.7) drops it before refinement.
OPENCV_FLOOR = dict(dog=dict(xc=30.4, yc=24.0, zc=1.0, A=3.5, a=12.0, c=4.0), gauss=dict(xc=30.4, yc=24.0, flat_radius=3.0))


def load_into_oracle(orc, case):
    """orc: an Oracle that has run on a W x H image with params_kw(); overwrite its planes, redo the keypoint stages."""
    for l, p in enumerate(dog_planes(**case["dog"])):
        orc.plane(0, 1, l, copy=False)[:] = p
    for l, p in enumerate(gauss_planes(**case["gauss"])):
        orc.plane(0, 0, l, copy=False)[:] = p
    return orc.run_keypoint_stages()
