#!/bin/bash
# GPU box: time the matcher for every build_variants/vN.so
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/popsift_amd/libpopsift_hip.so /tmp/orig.so
for so in $R/build_variants/v*.so; do
  n=$(basename $so .so); cp $so $R/popsift_amd/libpopsift_hip.so
  echo "== $(grep "^$n:" $R/build_variants/flags.txt)"
  (cd $R && python tools/match_time.py 95000 2>&1 | head -1; python -m pytest tests/test_gpu_match.py -x -q 2>&1 | tail -1)
done
cp /tmp/orig.so $R/popsift_amd/libpopsift_hip.so
