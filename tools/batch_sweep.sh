#!/bin/bash
# GPU box: throughput of the timed loop over (contexts, images per launch).  usage: batch_sweep.sh "C:B C:B ..."
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cb in ${1:-16:1 8:2 4:4 2:8 4:8 8:4}; do
  C=${cb%%:*}; B=${cb##*:}
  echo "== contexts $C, images per launch $B: $(timeout -k 10 300 python3 bench.py --quick --steps 8 --warmup 2 --contexts $C --launch-batch $B $EXTRA 2>&1 | tail -1)"
done
