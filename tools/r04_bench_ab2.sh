#!/bin/bash
# GPU box: bench.py --quick (dense + sparse) for the product library and every build_variants/vN.so, under the debug switches given.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for so in product $R/build_variants/v*.so; do
  if [ $so = product ]; then unset POPSIFT_HIP_LIB; n=main; else [ -f $so ] || continue; export POPSIFT_HIP_LIB=$so; n=$(basename $so .so); echo "## $(grep "^$n:" $R/build_variants/flags.txt)"; fi
  for dbg in "$@"; do
    [ "$dbg" = "-" ] && dbg=""
    for thr in "" "--threshold 0.17"; do
      out=$(timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 ${dbg:+--debug $dbg} $thr 2>/dev/null | tail -1)
      echo "$n debug='$dbg' $thr: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"])')"
    done
  done
done
