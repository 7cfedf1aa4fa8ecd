#!/bin/bash
# GPU box: waves per image in the launches of k_orientation / k_descriptor (popsift_hip_debug_set KP_WAVES = 7):
# kernel time per batch of 8 images (rocprofv3) and the quick bench, product library (or POPSIFT_HIP_LIB).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for n in ${@:-8192 16384 32768 65536 131072}; do
  echo "== KP_WAVES $n"
  PROF_DEBUG=7:$n timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kpw -- python3 $R/tools/prof_batch.py 8 3 > /tmp/kpw.log 2>&1 || exit 1
  python3 $R/tools/kstats.py /tmp/kpw | grep -E "k_descriptor|k_orientation"; rm -rf /tmp/kpw
  (cd $R && timeout -k 10 300 python3 bench.py --quick --steps 8 --warmup 2 --debug 7:$n 2>&1 | tail -1)
done
