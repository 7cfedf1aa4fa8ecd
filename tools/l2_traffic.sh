#!/bin/bash
# GPU box: L1 -> L2 request stream and L2 hit rate of every kernel in a batched launch (8 images), per image
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/l2_traffic.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/l2t -- python3 $R/tools/prof_batch.py 8 3 > /tmp/l2t.log 2>&1
  rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $R/tools/pmc.py /tmp/l2t >> $OUT; rm -rf /tmp/l2t
done
cat $OUT
