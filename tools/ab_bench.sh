#!/bin/bash
# GPU box: alternate the product library and build_variants/v1.so in the quick bench (same box, interleaved runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  unset POPSIFT_HIP_LIB
  echo "product: $(timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 2>&1 | tail -1)"
  export POPSIFT_HIP_LIB=$R/build_variants/v1.so
  echo "v1     : $(timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 2>&1 | tail -1)"
done
