#!/bin/bash
# GPU box: product + every build_variants/vN.so: k_descriptor per image in batched launches (rocprofv3 stats of
# tools/prof_batch.py 8 3) and the quick bench
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
one() {
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abv -- python3 $R/tools/prof_batch.py 8 3 > /tmp/abv.log 2>&1 || exit 1
  python3 $R/tools/kstats.py /tmp/abv | grep -E "k_descriptor|k_orientation"; rm -rf /tmp/abv
  (cd $R && timeout -k 10 300 python3 bench.py --quick --steps 8 --warmup 2 2>&1 | tail -1)
}
echo "== product"; unset POPSIFT_HIP_LIB; one
for so in $R/build_variants/v*.so; do
  n=$(basename $so .so); export POPSIFT_HIP_LIB=$so
  echo "== $(grep "^$n:" $R/build_variants/flags.txt)"; one
done
