#!/bin/bash
# GPU box: own times of the blur kernels at 8 images per launch (one context), product library and every build_variants/vN.so.
R=${GRAFT_REPO_ROOT:-/root/repo}
for so in product $R/build_variants/v*.so; do
  if [ $so = product ]; then unset POPSIFT_HIP_LIB; n=main; else [ -f $so ] || continue; export POPSIFT_HIP_LIB=$so; n=$(basename $so .so); echo "## $(grep "^$n:" $R/build_variants/flags.txt)"; fi
  $R/tools/r04_bench_kstats.sh solo_$n - --contexts 1 --launch-batch 8 2>/dev/null | grep -E "value|march|blur_tile<5, 1"
done
