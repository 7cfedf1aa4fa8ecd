#!/bin/bash
# GPU box: k_pyr_tail duration of one config-2 image for the product library and every build_variants/vN.so
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/tailv; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in product $R/build_variants/v*.so; do
  if [ $so = product ]; then unset POPSIFT_HIP_LIB; n=main; else [ -f $so ] || continue; export POPSIFT_HIP_LIB=$so; n=$(basename $so .so); fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$n -- python3 $R/tools/prof_run.py 5 > $OUT/$n.log 2>&1
  echo "$n $(grep "^$n:" $R/build_variants/flags.txt): $(python3 $R/tools/kstats.py $OUT/p_$n | grep tail)"
  rm -rf $OUT/p_$n
done
