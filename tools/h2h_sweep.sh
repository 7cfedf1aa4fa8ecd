#!/bin/bash
# host-to-host throughput of the C++ API (popsift-bench) over contexts per GPU; run on the GPU box from the repo root.
# usage: tools/h2h_sweep.sh [images]   -> gpurun_out/h2h_sweep.txt
R=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-96}
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
: > $R/gpurun_out/h2h_sweep.txt
for c in 1 2 3 4 6 8; do
  POPSIFT_CONTEXTS_PER_DEVICE=$c POPSIFT_PINNED_CACHE_MB=6000 $R/popsift_amd/popsift-bench --images $N --inflight $((c*4)) --pgm $FILES >> $R/gpurun_out/h2h_sweep.txt || exit 1
done
rm -rf $R/gpurun_out/pgm
cat $R/gpurun_out/h2h_sweep.txt
