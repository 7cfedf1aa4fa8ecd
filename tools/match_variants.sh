#!/bin/bash
# GPU box: time the matcher for every build_variants/vN.so (loaded through POPSIFT_HIP_LIB; the product library stays)
R=${GRAFT_REPO_ROOT:-/root/repo}
for so in $R/build_variants/v*.so; do
  n=$(basename $so .so); export POPSIFT_HIP_LIB=$so
  echo "== $(grep "^$n:" $R/build_variants/flags.txt)"
  (cd $R && python tools/match_time.py 95000 2>&1 | head -1; python -m pytest tests/test_gpu_match.py -x -q 2>&1 | tail -1)
done
