"""Host-side cost of one submit (all launches of an image enqueued) against the device time of the image: if the two are
close, a context is launch-bound.  python3 tools/submit_cost.py  (GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from popsift_amd import _capi as hip
from popsift_amd.synth import synth

W, H = 1920, 1080
img = torch.from_numpy(synth(2, W, H)).cuda()
ctx = hip.Context()
for _ in range(3):
    ctx.submit_dev(img.data_ptr(), W, H, W)
    ctx.wait()
ts, td = [], []
for _ in range(20):
    t0 = time.perf_counter()
    ctx.submit_dev(img.data_ptr(), W, H, W)
    t1 = time.perf_counter()
    ctx.wait()
    ts.append((t1 - t0) * 1e3)
    td.append(ctx.report().ms_device)
print("submit host ms: median %.3f min %.3f   device ms: median %.3f" % (np.median(ts), min(ts), np.median(td)))
