#!/bin/bash
# GPU box: per-launch timeline of the pyramid of one image for the two launch orders (PYR_ORDER 0 / 1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for ord in 0 1; do
  export PROF_DEBUG="6:$ord"
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/po_$ord -- python3 $R/tools/prof_run.py 5 > /tmp/po_$ord.log 2>&1 || exit 1
  echo "== PYR_ORDER $ord: $(grep -E 'ms$' /tmp/po_$ord.log | tail -1)"
  python3 $R/tools/ktrace.py /tmp/po_$ord 5 | sed -n 1,14p
  python3 $R/tools/ktrace.py /tmp/po_$ord 5 | tail -1
done
