#!/bin/bash
# GPU box: the GPU test suite, then single-image kernel times and the quick throughput bench of the product library
# (and of every build_variants/vN.so when called with "variants").
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/check
mkdir -p $OUT
cd $R
if [ "$1" != "notest" ]; then
  timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1
  rc=$?
  tail -15 $OUT/pytest.log
  # a timed-out or killed test run says something about the GPU: stop here; a failed assertion does not
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -ge 128 ]; then exit $rc; fi
  TEST_RC=$rc
fi
cd /tmp && export TMPDIR=/tmp
run_one() {
  n=$1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/chk_$n -- python3 $R/tools/prof_run.py 5 > $OUT/$n.log 2>&1 || exit 1
  python3 $R/tools/kstats.py /tmp/chk_$n > $OUT/$n.stats.txt
  grep -E "k_descriptor|k_orientation|k_scan|k_refine|k_detect<0, 3, false|k_blur_tile<5, 1" $OUT/$n.stats.txt
  grep -E "ms$" $OUT/$n.log | tail -1
  rm -rf /tmp/chk_$n
  (cd $R && timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 2>&1 | tail -1)
}
echo "== product (pytest rc=${TEST_RC:-skipped})"; unset POPSIFT_HIP_LIB; run_one main
if [ "$1" = "variants" ] || [ "$2" = "variants" ]; then
  for so in $R/build_variants/v*.so; do
    n=$(basename $so .so)
    export POPSIFT_HIP_LIB=$so
    echo "== $(grep "^$n:" $R/build_variants/flags.txt)"
    run_one $n
  done
fi
exit ${TEST_RC:-0}
