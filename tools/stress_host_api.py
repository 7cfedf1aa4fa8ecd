"""Stress of the pipelined worker loop of the C++ layer: the 72-image warped stream (mixed content, one size) and a
mixed-size batch through tests/cpp/host_batch_test with 1 / 2 / 3 / 6 contexts per GPU, several rounds each -- every
job's order-independent digest must be the same in every run.  python3 tools/stress_host_api.py [rounds]  (GPU box)"""
import os
import sys
import tempfile
import pathlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_configs45 as T
from popsift_amd.synth import oxford_like_stream, synth

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T._build()
stream = [im for _, _, im in oxford_like_stream(seeds=range(200, 206))]
mixed = [synth(300 + k, 200 + 37 * (k % 9), 150 + 23 * (k % 7)) for k in range(48)]
bad = 0
for name, imgs in (("warped 800x640 x %d" % len(stream), stream), ("mixed sizes x %d" % len(mixed), mixed)):
    want = None
    for per in (1, 2, 3, 6):
        for r in range(rounds):
            with tempfile.TemporaryDirectory() as td:
                rows = T._run_batch(pathlib.Path(td), imgs, env={"POPSIFT_CONTEXTS_PER_DEVICE": str(per)})
            key = [(nf, nd, dg) for _, nf, nd, dg in rows]
            if want is None:
                want = key
            ok = key == want
            bad += 0 if ok else 1
            print(name, "contexts", per, "round", r, "ok" if ok else "DIFFERENT", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
