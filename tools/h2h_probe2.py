"""Is popsift-bench slower as a child of a process that has initialised torch / HIP?  (bench.py's C++ leg)"""
import os, subprocess, sys, json, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from popsift_amd.synth import synth
os.makedirs("/tmp/h2h", exist_ok=True)
pg = []
for k in range(8):
    im = synth(2 if k == 0 else 100 + k, 1920, 1080)
    pg.append("/tmp/h2h/img%d.pgm" % k)
    with open(pg[-1], "wb") as f:
        f.write(b"P5\n1920 1080\n255\n"); f.write(im.tobytes())
def leg(tag, extra_env=None):
    env = dict(os.environ, POPSIFT_CONTEXTS_PER_DEVICE="4", POPSIFT_DEVICES="0", POPSIFT_BATCH="1", POPSIFT_PINNED_CACHE_MB="2800")
    env.update(extra_env or {})
    r = subprocess.run([os.path.join(R, "popsift_amd", "popsift-bench"), "--images", "64", "--inflight", "16", "--callers", "2",
                        "--pgm", ",".join(pg)], capture_output=True, text=True, env=env)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(tag, d["e2e_host_api_mpix_s"], d["caller_us_per_image"], flush=True)
leg("before torch")
import torch
leg("torch imported")
torch.cuda.init(); x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
leg("torch cuda initialised")
leg("... again")
big = torch.zeros(3 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); del big; torch.cuda.empty_cache()
leg("after a 3 GiB tensor came and went")
print("affinity", len(os.sched_getaffinity(0)), "threads", torch.get_num_threads())
leg("OMP/MKL threads = 1", {"OMP_NUM_THREADS": "1"})
# ... and after this process has itself extracted on the GPU, as bench.py has when it starts its C++ leg
from popsift_amd import _capi as hip
img = synth(2, 1920, 1080)
c = hip.Context(hip.default_params()); c.submit(img); c.wait(); c.close()
leg("after one context here came and went")
cs = [hip.Context(hip.default_params()) for _ in range(3)]
for c in cs:
    c.submit_batch([img] * 8); c.wait_batch()
leg("with three 8-slot contexts alive here")
for c in cs:
    c.close()
leg("after they were closed")
import gc; gc.collect(); torch.cuda.empty_cache()
leg("after gc")
