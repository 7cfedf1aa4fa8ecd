#!/bin/bash
# GPU box: bench.py --quick under several debug switch settings (dense and keypoint-sparse).  tools/r04_bench_ab.sh "<dbg1>" "<dbg2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for dbg in "$@"; do
  [ "$dbg" = "-" ] && dbg=""
  for thr in "" "--threshold 0.17"; do
    out=$(timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 ${dbg:+--debug $dbg} $thr 2>/dev/null | tail -1)
    echo "debug='$dbg' $thr: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"])')"
  done
done
