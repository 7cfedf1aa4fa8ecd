#!/bin/bash
# Runs on the GPU box (via gpurun): SQ / TCC counters of the keypoint kernels (k_orientation, k_descriptor, k_refine,
# k_detect) for one context extracting 3 config-2 images.  One rocprofv3 --pmc pass per counter group (8 SQ slots,
# 4 TCC slots per pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"); --kernel-trace only, no other trace domain.
#   usage: collect_kp_counters.sh <tag>      -> gpurun_out/kpc_<tag>/<group>/...csv + gpurun_out/kpc_<tag>/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-base}
OUT=$R/gpurun_out/kpc_$TAG
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1
run_pass() {
    name=$1
    shift
    timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $R/tools/prof_run.py 3 > $OUT/$name.log 2>&1
    rc=$?
    echo "pass $name rc=$rc" >> $OUT/passes.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "pass $name timed out: stopping" >> $OUT/passes.txt
        exit 1
    fi
}
run_pass sq_inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
run_pass sq_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_LDS_ATOMIC_RETURN SQ_INSTS_FLAT
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run_pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
for p in sq_inst sq_wait sq_lds tcc tcp grbm fetch write; do
    echo "== $p" >> $OUT/summary.txt
    python3 $R/tools/pmc.py $OUT/$p >> $OUT/summary.txt 2>&1
done
cat $OUT/passes.txt
# the raw per-dispatch CSVs are tens of MB (gpurun merges at most 64 MiB back): keep the summary and the logs
for p in sq_inst sq_wait sq_lds tcc tcp grbm fetch write; do rm -rf $OUT/$p; done
