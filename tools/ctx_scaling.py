"""Device-resident input, N contexts in N threads, WITHOUT torch in the process (HIP runtime = /opt/rocm's).
bench.py imports torch first, whose bundled HIP runtime then serves libpopsift_hip.so as well."""
import os, sys, time, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.init()
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth

img = synth(2, 1920, 1080)
L = hip.lib()
rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
dev = C.c_void_p()
assert rt.hipMalloc(C.byref(dev), img.nbytes) == 0
assert rt.hipMemcpy(dev, img.ctypes.data, img.nbytes, 1) == 0
v = C.c_int(); rt.hipRuntimeGetVersion(C.byref(v)); print("hip runtime version", v.value, flush=True)

def worker(ctx, n):
    for _ in range(n):
        ctx.submit_dev(dev, 1920, 1080, 1920)
        ctx.wait()

for nctx in (1, 2, 4, 8, 16):
    ctxs = [hip.Context() for _ in range(nctx)]
    for c in ctxs:
        c.submit_dev(dev, 1920, 1080, 1920); c.wait()
    per = max(48 // nctx, 3)
    ths = [threading.Thread(target=worker, args=(c, per)) for c in ctxs]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    n = per * nctx
    print("contexts %2d: %.3f ms per image  %.1f Mpix/s" % (nctx, dt * 1e3 / n, n * img.size / dt / 1e6), flush=True)
    for c in ctxs: c.close()
