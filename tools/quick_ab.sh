#!/bin/bash
# GPU box: single-image kernel times of the product library + every build_variants/vN.so, then the throughput bench
# (quick mode) for each of them and, for the product library, with the spatial ordering switched off.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/quick_ab
rm -rf $OUT; mkdir -p $OUT
PAT=${1:-k_descriptor|k_orientation|k_scan_local|k_order}
cd /tmp && export TMPDIR=/tmp
run_one() {
  n=$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -- python3 $R/tools/prof_run.py 5 > $OUT/$n.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $OUT/$n > $OUT/$n.stats.txt
  grep -E "$PAT" $OUT/$n.stats.txt
  grep -E "ms$" $OUT/$n.log | tail -1
  rm -rf $OUT/$n
  (cd $R && timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 2>&1 | tail -1)
}
echo "== product"; unset POPSIFT_HIP_LIB; run_one main
for so in $R/build_variants/v*.so; do
  n=$(basename $so .so)
  export POPSIFT_HIP_LIB=$so
  echo "== $(grep "^$n:" $R/build_variants/flags.txt)"
  run_one $n
done
