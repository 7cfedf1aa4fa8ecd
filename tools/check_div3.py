"""x / 3.0f by multiplication (keypoint.hip div3): q = x * RN(1/3); q + fma(-3, q, x) * RN(1/3) must equal the
correctly rounded quotient.  Exhaustive over all 2^23 floats of six binades (the pattern repeats per binade away
from under- / overflow).  Runs on the CPU in about a minute."""
import numpy as np

r = np.float32(1.0) / np.float32(3.0)


def fma32(a, b, c):  # longdouble holds the 48-bit product exactly; one rounding to float at the end
    return (a.astype(np.longdouble) * b.astype(np.longdouble) + c.astype(np.longdouble)).astype(np.float32)


bad = 0
for e in (0x3F800000, 0x40000000, 0x47000000, 0x2F000000, 0x01000000, 0x7E800000):
    x = (np.arange(1 << 23, dtype=np.uint32) + np.uint32(e)).view(np.float32)
    q = x * r
    q1 = fma32(fma32(np.full_like(q, -3.0), q, x), np.full_like(q, r), q)
    ref = (x.astype(np.longdouble) / np.longdouble(3)).astype(np.float32)
    n = int((q1.view(np.uint32) != ref.view(np.uint32)).sum())
    print(hex(e), "mismatches", n)
    bad += n
print("OK" if bad == 0 else "FAILED")
