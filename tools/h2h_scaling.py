"""Host-to-host throughput through the C ABI with N contexts in N threads (pinned result buffers)."""
import os, sys, time, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth

img = synth(2, 1920, 1080)
L = hip.lib()
FETCH = os.environ.get("H2H_FETCH", "1") == "1"

def pinned(nbytes, dtype, shape):
    p = L.popsift_hip_host_alloc(nbytes)
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_ubyte)), (nbytes,)).view(dtype).reshape(shape)

def worker(ctx, n, out, feats, desc):
    for _ in range(n):
        ctx.submit(img)
        nf, nd = ctx.wait()
        if FETCH:
            rc = L.popsift_hip_fetch(ctx._h, feats.ctypes.data, 100000, desc.ctypes.data, 130000 * 128)
            assert rc == 0
    out.append(1)

for nctx in (1, 2, 4, 8, 16):
    ctxs = [hip.Context() for _ in range(nctx)]
    for c in ctxs:
        c.submit(img); c.wait()
    per = max(48 // nctx, 3)
    out = []
    bufs = [(pinned(100000 * hip.FEATURE_DTYPE.itemsize, hip.FEATURE_DTYPE, (100000,)),
             pinned(130000 * 512, np.float32, (130000, 128))) for _ in ctxs]      # outside the timed region
    ths = [threading.Thread(target=worker, args=(c, per, out, b[0], b[1])) for c, b in zip(ctxs, bufs)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    n = per * nctx
    print("contexts %d fetch %d: %.3f ms per image  %.1f Mpix/s" % (nctx, FETCH, dt * 1e3 / n, n * img.size / dt / 1e6), flush=True)
    for c in ctxs: c.close()
