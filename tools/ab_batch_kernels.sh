#!/bin/bash
# GPU box: kernel durations of batched launches (B images per launch), product library vs build_variants/v1.so
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-8}
cd /tmp && export TMPDIR=/tmp
for v in product v1; do
  unset POPSIFT_HIP_LIB; [ $v = v1 ] && export POPSIFT_HIP_LIB=$R/build_variants/v1.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$v -- python3 $R/tools/prof_batch.py $B 3 > /tmp/pb_$v.log 2>&1 || exit 1
  echo "== $v (B=$B): $(tail -1 /tmp/pb_$v.log)"
  python3 $R/tools/kstats.py /tmp/pb_$v | head -14
done
