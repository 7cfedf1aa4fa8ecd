#!/bin/bash
# GPU box: bench.py --quick over contexts x images-per-launch (dense and keypoint-sparse).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "$@"; do
  c=${cfg%x*}; lb=${cfg#*x}
  for thr in "" "--threshold 0.17"; do
    out=$(timeout -k 10 300 python3 bench.py --quick --steps 8 --warmup 2 --contexts $c --launch-batch $lb $thr 2>/dev/null | tail -1)
    echo "contexts $c x $lb images/launch $thr: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"])')"
  done
done
