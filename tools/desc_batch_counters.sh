#!/bin/bash
# GPU box: k_descriptor in a batched launch (8 images, planes far beyond the caches) against the single-image launch:
# wave-cycle breakdown, memory and LDS latency, L2 hit rate.  Three rocprofv3 --pmc passes each.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/desc_batch_counters.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; prog=$1; shift
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
    timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/dbc -- python3 $R/tools/$prog "$@" > /tmp/dbc.log 2>&1
    rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    echo "== $tag" >> $OUT
    python3 $R/tools/pmc.py /tmp/dbc | grep -E "^kernel|k_descriptor|k_orientation" >> $OUT; rm -rf /tmp/dbc
  done; }
run "single image x3" prof_run.py 3
run "batch of 8 x3" prof_batch.py 8 3
cat $OUT
