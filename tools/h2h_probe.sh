#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from popsift_amd.synth import synth
os.makedirs("/tmp/h2h", exist_ok=True)
for k in range(8):
    im = synth(2 if k == 0 else 100 + k, 1920, 1080)
    with open("/tmp/h2h/img%d.pgm" % k, "wb") as f:
        f.write(b"P5\n1920 1080\n255\n"); f.write(im.tobytes())
PY
PGMS=$(ls /tmp/h2h/img*.pgm | tr '\n' ',' | sed 's/,$//')
run() { echo "$1: $(env $2 timeout -k 10 200 ./popsift_amd/popsift-bench $3 --pgm $PGMS 2>&1 | tail -1 | cut -c1-330)"; }
run "bench.py form (64 img, inflight 16, cache 2800)" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_BATCH=1 POPSIFT_PINNED_CACHE_MB=2800 POPSIFT_DEVICES=0" "--images 64 --inflight 16 --callers 2"
run "128 img" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_BATCH=1 POPSIFT_PINNED_CACHE_MB=2800 POPSIFT_DEVICES=0" "--images 128 --inflight 16 --callers 2"
run "inflight 12" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_BATCH=1 POPSIFT_PINNED_CACHE_MB=2800 POPSIFT_DEVICES=0" "--images 128 --inflight 12 --callers 2"
run "cache 4000" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_BATCH=1 POPSIFT_PINNED_CACHE_MB=4000 POPSIFT_DEVICES=0" "--images 128 --inflight 12 --callers 2"
run "no DEVICES env" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_BATCH=1 POPSIFT_PINNED_CACHE_MB=4000" "--images 128 --inflight 12 --callers 2"
run "callers 1" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_BATCH=1 POPSIFT_PINNED_CACHE_MB=4000" "--images 128 --inflight 12 --callers 1"
run "numa off" "POPSIFT_CONTEXTS_PER_DEVICE=4 POPSIFT_NUMA_BIND=0 POPSIFT_PINNED_CACHE_MB=4000" "--images 128 --inflight 12 --callers 2"
