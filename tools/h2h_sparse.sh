#!/bin/bash
# host-to-host throughput of the C++ API in the keypoint-sparse regime (threshold 0.17) over contexts per GPU; GPU box.
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out/pgm
python3 - <<PY
import sys
sys.path.insert(0, "$R")
from popsift_amd.synth import synth
for s in range(100, 108):
    im = synth(s, 1920, 1080)
    with open("$R/gpurun_out/pgm/s%d.pgm" % s, "wb") as f:
        f.write(b"P5\n1920 1080\n255\n" + im.tobytes())
PY
FILES=$(ls $R/gpurun_out/pgm/*.pgm | paste -sd, -)
for c in 1 2 3 4 6 8; do
  POPSIFT_CONTEXTS_PER_DEVICE=$c $R/popsift_amd/popsift-bench --images 256 --inflight $((c*4)) --threshold 0.17 --pgm $FILES || exit 1
done
rm -rf $R/gpurun_out/pgm
