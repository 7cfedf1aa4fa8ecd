"""Throughput of the keypoint-sparse regime (threshold 0.17: 2 features / 1000 px) over the number of contexts in flight.
python3 tools/sparse_throughput.py [idle]  (GPU box; `idle`: with 16 idle default contexts alive, as inside bench.py --
the runtime deals its hardware queues to streams in creation order, so streams that merely exist change the result)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from popsift_amd import _capi as hip
from popsift_amd.synth import synth

W, H = 1920, 1080
bench.W, bench.H = W, H
imgs = [torch.from_numpy(synth(100 + k, W, H)).cuda() for k in range(16)]
ptrs = [imgs[i % 16].data_ptr() for i in range(64)]
idle = [hip.Context() for _ in range(16)] if len(sys.argv) > 1 else []
for n in (2, 4, 8, 12, 16, 24):
    ctxs = [hip.Context(hip.default_params(threshold=0.17)) for _ in range(n)]
    w = bench.Workers(ctxs, ptrs)
    w.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        w.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    w.close()
    for c in ctxs:
        c.close()
    print("contexts %2d: %.0f Mpix/s, %.3f ms per image" % (n, 4 * 64 * W * H / 1e6 / dt, dt / 256 * 1e3), flush=True)
