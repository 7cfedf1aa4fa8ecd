"""Profiling driver: N submits of a batch of B config-2 images on one context (kernel durations per launch / B = per image
in a batched launch).  usage: prof_batch.py [B] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popsift_amd import _capi as hip
from popsift_amd.synth import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
imgs = [synth(2 if k == 0 else 100 + k, 1920, 1080) for k in range(min(B, 4))]
imgs = [imgs[k % len(imgs)] for k in range(B)]
ctx = hip.Context(hip.default_params())
for item in filter(None, os.environ.get("PROF_DEBUG", "").split(",")):   # popsift_hip_debug_set switches "what:value,..."
    ctx.debug_set(int(item.split(":")[0]), int(item.split(":")[1]))
for i in range(N):
    ctx.submit_batch(imgs)
    c = ctx.wait_batch()
    print(len(c), c[0], "%.3f ms per image" % (ctx.report().ms_device / B), flush=True)
