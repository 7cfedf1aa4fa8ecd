#!/bin/bash
# Runs on the GPU box (via gpurun): collects the rocprofv3 evidence that profiles/ summarises.
#   1. kernel trace + stats of the bench command itself
#   2. kernel trace + stats of one context extracting 5 images back to back (per-launch timeline)
#   3. HBM traffic of the kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
#      (MI355X_MICROARCH.md "HBM": TCC slots; FETCH_SIZE counts half the bytes of wide reads)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profiles_raw
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roofline -- python3 $R/bench.py --only-roofline > $OUT/roofline.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single -- python3 $R/tools/prof_run.py 5 > $OUT/single.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/prof_run.py 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/prof_run.py 3 > $OUT/pmc_write.log 2>&1 || exit 1
tail -1 $OUT/bench.log | cut -c1-400
