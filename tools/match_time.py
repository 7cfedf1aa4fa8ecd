"""Wall time of popsift_hip_match_sets (allocations, both kernels, result download) for a few set sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from popsift_amd import _capi as hip

rng = np.random.default_rng(0)
SIZES = ((1000, 1000), (5000, 5000), (20000, 20000), (95000, 95000))
if len(sys.argv) > 1:
    SIZES = ((int(sys.argv[1]), int(sys.argv[1])),)
for nl, nr in SIZES:
    l = rng.random((nl, 128), np.float32); l /= np.linalg.norm(l, axis=1, keepdims=True)
    r = rng.random((nr, 128), np.float32); r /= np.linalg.norm(r, axis=1, keepdims=True)
    L, R = hip.DevFeatures.from_host(l), hip.DevFeatures.from_host(r)
    L.match(R)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m = L.match(R); ts.append(time.perf_counter() - t0)
    t = min(ts)
    pairs = nl * nr
    print("match %6d x %6d: %8.3f ms  %7.1f Gpairs/s  %6.1f TFLOP/s (3 flop x 128 per pair)  accept %.3f" % (
        nl, nr, t * 1e3, pairs / t / 1e9, pairs * 384 / t / 1e12, m["accept"].mean()), flush=True)

# real descriptors: two unrelated 1080p synthetic images (no true correspondences: the hardest case for screening)
from popsift_amd.synth import synth
a = hip.Context().submit(synth(2, 1920, 1080))
b = hip.Context().submit(synth(102, 1920, 1080))
A, B = a.clone_results(), b.clone_results()
A.match(B)
ts = []
for _ in range(3):
    t0 = time.perf_counter(); m = A.match(B); ts.append(time.perf_counter() - t0)
nl, nr = A.info()[2], B.info()[2]
print("match %6d x %6d SIFT descriptors of two images: %8.3f ms  %7.1f Gpairs/s  accept %.3f" % (
    nl, nr, min(ts) * 1e3, nl * nr / min(ts) / 1e9, m["accept"].mean()), flush=True)
