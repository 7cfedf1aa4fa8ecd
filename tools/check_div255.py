#!/usr/bin/env python3
"""v / 255.0f for v = 0 .. 255 (cudaReadModeNormalizedFloat, s_image.cu:140-169) as multiply + two FMAs, in EXACT rational
arithmetic: q = RN(v * r), rem = RN(v - 255 q) (exact), result = RN(rem * r + q) with r = RN(1 / 255) must equal
RN(v / 255) for all 256 values -- the level-0 kernel (pyramid.hip, fast2x == 2) converts its texels that way."""
from fractions import Fraction

import numpy as np


def rn32(fr):
    x = np.float32(float(fr))
    best = None
    for c in (np.nextafter(x, np.float32(-np.inf)), x, np.nextafter(x, np.float32(np.inf))):
        d = abs(Fraction(float(c)) - fr)
        even = (np.float32(c).view(np.uint32) & 1) == 0
        if best is None or d < best[0] or (d == best[0] and even):
            best = (d, np.float32(c))
    return best[1]


r = np.float32(1.0) / np.float32(255.0)
bad = 0
for v in range(256):
    ref = rn32(Fraction(v, 255))
    q = rn32(Fraction(v) * Fraction(float(r)))
    rem = rn32(Fraction(v) - 255 * Fraction(float(q)))
    bad += int(rn32(Fraction(float(rem)) * Fraction(float(r)) + Fraction(float(q))) != ref)
print("r = %r, mismatches: %d of 256" % (float(r), bad))
raise SystemExit(1 if bad else 0)
