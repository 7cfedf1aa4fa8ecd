import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from popsift_amd import _capi as hip
from oracle import oracle as O
from test_golden import load_case
z, kw = load_case("tests/golden/noupscale_levels4_150x110.npz")
print(kw)
img = z["image"]
orc = O.Oracle(O.default_params(**kw), threads=4).run(img)
ctx = hip.Context(hip.default_params(**kw)); ctx.submit(img); ctx.wait()
fo, so, go = orc.gauss_table(); fh, sh, gh = ctx.gauss_table()
print("tables equal", np.array_equal(fo, fh), so, sh)
for l in (0, 3):
    a = orc.plane(0, 0, l); b = ctx.plane(0, 0, l); g = z["g_o0_l%d" % l]
    print("level", l, "oracle-vs-hip diff px", int((a != b).sum()), "oracle-vs-golden", int((a[::4, ::4] != g).sum()), "hip-vs-golden", int((b[::4, ::4] != g).sum()))
    ys, xs = np.nonzero(a != b)
    print("  where", list(zip(xs[:10].tolist(), ys[:10].tolist())), a[a != b][:5], b[a != b][:5])
