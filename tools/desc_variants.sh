#!/bin/bash
# GPU box: per-kernel times of every build_variants/vN.so (single image, config 2) + smoke check.
# The product library is never touched: the Python binding loads the variant through POPSIFT_HIP_LIB.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/variants
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PAT=${1:-k_descriptor|k_orientation}
for so in $R/build_variants/v*.so; do
  n=$(basename $so .so)
  export POPSIFT_HIP_LIB=$so
  echo "== $(grep "^$n:" $R/build_variants/flags.txt)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -- python3 $R/tools/prof_run.py 5 > $OUT/$n.log 2>&1 || exit 1
  python3 $R/tools/kstats.py $OUT/$n | grep -E "$PAT"
  grep -E "ms$" $OUT/$n.log | tail -1
  (cd $R && timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1)
done
