"""GPU box, diagnostic build only (tools/experiments/r04_march_stamps.patch applied, -DMARCH_STAMPS, loaded through POPSIFT_HIP_LIB): where the waves of k_blur_march<5> spend
their cycles on one config-2 image, from s_memtime stamps at the phase boundaries.
    python3 tools/r04_march_stamps.py <seg_rows> [n_cus_to_print]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
seg = int(sys.argv[1]); ncu = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # images per launch (the stamps are image 0's)
PH, N = 6, 8
img = synth(2, 1920, 1080)
imgs = [synth(2 + k, 1920, 1080) for k in range(nb)] if nb > 1 else [img]
ctx = hip.Context()
ctx.debug_set(hip.DEBUG_BLUR_PATH, 2); ctx.debug_set(hip.DEBUG_BLUR_SEG, seg); ctx.debug_set(hip.DEBUG_PYR_TAIL, 1)
for _ in range(4):
    if nb == 1:
        ctx.submit(img); ctx.wait()
    else:
        ctx.submit_batch([synth(2 + k, 1920, 1080) if _ == 0 else imgs[k] for k in range(nb)] if False else imgs); ctx.wait_batch()
L = hip.lib()
buf = np.zeros(4096 * 4 * PH * N, np.uint64)
L.popsift_hip_debug_read_march_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.popsift_hip_debug_read_march_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(4096, 4, PH, N).astype(np.int64)
strips = 30; segs = (2160 + seg - 1) // seg; nwg = strips * segs
st = st[:nwg]
t0 = st[:, :, 0, 0].min()
K = min((min(seg, 2160) + 31) // 32, PH - 1)
print("seg_rows %d: %d workgroups, %d steps each (stamped phases 0..%d)" % (seg, nwg, (min(seg, 2160) + 31) // 32, K))
hw = st[:, 0, PH - 1, 7]
cu_key = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xff) | (((hw & 0xffffffff) >> 13 & 0x7) << 8)
names = ["wait+stage (t1->t2)", "stores+loads issue+H pass (t2->t3)", "barrier b (t3->t4)", "V pass (t4->t5)"]
for p in range(0, K + 1):
    seg_d = []
    for a, b in ((1, 2), (2, 3), (3, 4), (4, 5)):
        if p == 0 and a >= 3: seg_d.append(float('nan')); continue
        d = st[:, :, p, b] - st[:, :, p, a]
        seg_d.append(float(np.median(d)))
    nxt = (st[:, :, p + 1, 1] - st[:, :, p, 1]) if p + 1 <= K else None
    print("phase %d: median cycles  %s | phase length %s" % (p, "  ".join("%s %6.0f" % (n.split(' (')[0], v) for n, v in zip(names, seg_d)),
                                                              "%.0f" % np.median(nxt) if nxt is not None else "-"))
life = st[:, :, PH - 1, 6] - st[:, :, 0, 0]
print("wave lifetime: median %.0f cycles, min %.0f, max %.0f; launch span %.0f cycles" % (np.median(life), life.min(), life.max(),
      st[:, :, PH - 1, 6].max() - t0))
# start times: how many rounds
starts = np.sort(st[:, 0, 0, 0] - t0)
print("workgroup start times (cycles): 10%% %.0f  50%% %.0f  75%% %.0f  90%% %.0f  max %.0f" % tuple(np.percentile(starts, [10, 50, 75, 90, 100])))
cus, counts = np.unique(cu_key, return_counts=True)
print("%d distinct CUs seen; workgroups per CU: min %d median %d max %d" % (len(cus), counts.min(), np.median(counts), counts.max()))
for cu in cus[:ncu]:
    print("CU key %x:" % cu)
    idx = np.nonzero(cu_key == cu)[0]
    for i in sorted(idx, key=lambda i: st[i, 0, 0, 0]):
        w = st[i, 0]
        ev = []
        for p in range(0, K + 1):
            ev.append("p%d[%d %d %d %d %d]" % (p, *(w[p, k] - t0 for k in (1, 2, 3, 4, 5))))
        print("   wg %4d start %6d end %6d  %s" % (i, w[0, 0] - t0, w[PH - 1, 6] - t0, " ".join(ev)))
ctx.close()
