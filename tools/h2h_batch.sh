#!/bin/bash
# GPU box: host image in -> host features out through the C++ API (popsift-bench) over POPSIFT_BATCH (jobs a worker takes
# per submit) x contexts per GPU, dense and keypoint-sparse.  usage: h2h_batch.sh   -> gpurun_out/h2h_batch.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/h2h_batch.txt
: > $OUT
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
for thr in 0.04 0.17; do
  for cb in 4:1 2:4 2:8 3:8 1:8 4:4; do
    C=${cb%%:*}; B=${cb##*:}
    n=128; [ "$thr" = "0.17" ] && n=512
    line=$(POPSIFT_CONTEXTS_PER_DEVICE=$C POPSIFT_BATCH=$B POPSIFT_PINNED_CACHE_MB=4000 timeout -k 10 200 ./popsift_amd/popsift-bench --images $n --inflight $((C * B + 8)) --callers 2 --threshold $thr --pgm $PGMS 2>&1 | tail -1)
    echo "threshold $thr contexts $C batch $B: $line" | tee -a $OUT
  done
done
