#!/bin/bash
# GPU box: where k_descriptor's waves wait -- average latency of its vector-memory and LDS instructions (LEVEL / count),
# FIFO back-pressure, instruction fetch.  Two rocprofv3 --pmc passes of tools/prof_run.py 3 (kernel trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/desc_latency
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/dl_$n -- python3 $R/tools/prof_run.py 3 > $OUT/$n.log 2>&1
  rc=$?; echo "pass $n rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  python3 $R/tools/pmc.py /tmp/dl_$n | grep -E "^kernel|k_descriptor|k_orientation|k_detect<0, 3, false|k_blur_tile<8, 0, 64" >> $OUT/summary.txt; rm -rf /tmp/dl_$n; }
pass mem SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass fifo SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_LDS_ATOMIC
cat $OUT/summary.txt
