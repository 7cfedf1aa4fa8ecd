#!/bin/bash
# GPU box: bench.py --quick with two builds of the library in turn (A B A B), dense and keypoint-sparse.
#   tools/r04_lib_ab.sh <libA.so> <libB.so> [rounds]   (paths relative to the repo root)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for lib in $A $B; do
    for thr in "" "--threshold 0.17"; do
      out=$(POPSIFT_HIP_LIB=$R/$lib timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 $thr 2>gpurun_out/ab_err.log | tail -1)
      echo "$lib $thr: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
    done
  done
done
