#!/bin/bash
# GPU box: kernel stats of the quick bench (one context, 16 images per launch: no overlap between kernels), product vs v1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in product v1; do
  unset POPSIFT_HIP_LIB; [ $v = v1 ] && export POPSIFT_HIP_LIB=$R/build_variants/v1.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk_$v -- python3 $R/bench.py --quick --steps 4 --warmup 1 --contexts ${1:-1} --launch-batch ${2:-16} > /tmp/bk_$v.log 2>&1 || exit 1
  echo "== $v: $(grep value /tmp/bk_$v.log | tail -1)"
  python3 $R/tools/kstats.py /tmp/bk_$v | head -12
done
