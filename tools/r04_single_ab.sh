#!/bin/bash
# GPU box: per-kernel times of single images (tools/prof_run.py 5, one context) with two builds of the library.
#   tools/r04_single_ab.sh <libA.so> <libB.so> <kernel name pattern>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in $1 $2; do
  rm -rf /tmp/sab; mkdir -p /tmp/sab
  POPSIFT_HIP_LIB=$R/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sab -- python3 $R/tools/prof_run.py 5 > /tmp/sab.log 2>&1 || { tail -5 /tmp/sab.log; exit 1; }
  echo "== $lib"
  python3 $R/tools/kstats.py /tmp/sab | grep -E "$3"
done
