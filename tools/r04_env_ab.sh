#!/bin/bash
# GPU box: single-image first-to-last span (rocprofv3 kernel trace, 5 images) under runtime environment settings.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/envab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  ( [ "$e" != "-" ] && export $e; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/prof_run.py 5 > $OUT/p$i.log 2>&1 )
  python3 $R/tools/ktrace.py $OUT/p$i 5 > $OUT/p$i.trace.txt
  echo "== $e: $(tail -1 $OUT/p$i.trace.txt) | small launches: $(grep -E 'k_blur_small|k_blur_duo<1[03], 4>' $OUT/p$i.trace.txt | awk '{s+=$4; n++} END {printf "%d launches avg %.2f us", n, s/n}') | $(grep ms $OUT/p$i.log | tail -1)"
  rm -rf $OUT/p$i
done
