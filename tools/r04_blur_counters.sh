#!/bin/bash
# GPU box: SQ counters of the blur kernels (one context, 3 config-2 images), one rocprofv3 --pmc pass per group.
#   tools/r04_blur_counters.sh <tag>   (POPSIFT_HIP_LIB / PROF_DEBUG are passed through)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-base}
OUT=$R/gpurun_out/blurc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pass() {
    name=$1; shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/tools/prof_run.py 3 > $OUT/$name.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then echo "pass $name rc=$rc"; tail -3 $OUT/$name.log; fi
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
    echo "== $name" >> $OUT/summary.txt
    python3 $R/tools/pmc.py $OUT/$name | grep -E "^kernel|blur|tail" >> $OUT/summary.txt 2>&1
    rm -rf $OUT/$name
}
run_pass sq_inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run_pass sq_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cat $OUT/summary.txt
