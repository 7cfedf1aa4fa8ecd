#!/bin/bash
# Runs on the GPU box (via gpurun): collects the rocprofv3 evidence that profiles/ summarises, and summarises it there
# (the raw per-dispatch CSVs are far beyond the 64 MiB gpurun merges back).
#   0. kernel trace + stats + who-runs-beside-whom timeline of the TIMED LOOP alone (bench.py --quick)
#   1. kernel trace + stats of the bench command itself
#   2. kernel trace + stats of one context extracting 5 images back to back (per-launch timeline)
#   3. counters of every kernel, one rocprofv3 --pmc pass per group (MI355X_MICROARCH.md "rocprofv3 PMC slots";
#      FETCH_SIZE and WRITE_SIZE in SEPARATE passes; --kernel-trace only, no other trace domain)
# usage: collect_profiles.sh <tag>   ->  gpurun_out/profiles_<tag>/*.txt, kernel_counters.json  (copy into profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
RAW=/tmp/profiles_raw
OUT=$R/gpurun_out/profiles_$TAG
rm -rf $RAW $OUT
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
NIMG=3
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/quick -- python3 $R/bench.py --quick --steps 6 --warmup 2 > $RAW/quick.log 2> $RAW/quick.err || exit 1
python3 $R/tools/mix_timeline.py $RAW/quick > $OUT/${TAG}_bench_quick_timeline.txt 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $RAW/bench.log 2> $RAW/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/single -- python3 $R/tools/prof_run.py 5 > $RAW/single.log 2>&1 || exit 1
pass() {
    name=$1
    shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $RAW/$name -- python3 $R/tools/prof_run.py $NIMG > $RAW/$name.log 2>&1
    rc=$?
    echo "pass $name rc=$rc" >> $OUT/passes.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
}
pass sq_inst SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass sq_wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
pass sq_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass grbm GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 $R/tools/summarize_profiles.py $TAG $RAW $OUT $NIMG
cat $OUT/passes.txt
ls -la $OUT
