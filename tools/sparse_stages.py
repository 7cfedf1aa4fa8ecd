"""Stage times (profile mode 2) of the keypoint-sparse leg of bench.py (threshold 0.17) next to the default image.
python3 tools/sparse_stages.py  (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from popsift_amd import _capi as hip
from popsift_amd.synth import synth

W, H = 1920, 1080
img = torch.from_numpy(synth(2, W, H)).cuda()
for name, kw in (("default", {}), ("sparse", dict(threshold=0.17))):
    ctx = hip.Context(hip.default_params(**kw))
    for _ in range(3):
        ctx.submit_dev(img.data_ptr(), W, H, W)
        ctx.wait()
    lat = []
    for _ in range(9):
        ctx.submit_dev(img.data_ptr(), W, H, W)
        ctx.wait()
        lat.append(ctx.report().ms_device)
    ctx.set_profile(2)
    st = []
    for _ in range(9):
        ctx.submit_dev(img.data_ptr(), W, H, W)
        ctx.wait()
        st.append(list(ctx.report().ms_stage)[:len(hip.STAGES)])
    r = ctx.report()
    print(name, "features", r.ext_total, "desc", r.ori_total, "T_dev %.3f ms" % np.median(lat),
          " ".join("%s %.1f" % (n, 1e3 * v) for n, v in zip(hip.STAGES, np.median(np.array(st), 0))), "(us)")
    ctx.close()
