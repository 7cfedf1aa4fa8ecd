#!/bin/bash
# GPU box: parity of the blur paths, then kernel times of one config-2 image (rocprofv3 kernel trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-march}
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_blur_paths.py -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/tools/prof_run.py 5 > $OUT/prof.log 2>&1 || { tail -20 $OUT/prof.log; exit 1; }
python3 $R/tools/kstats.py $OUT/prof > $OUT/kstats.txt
python3 $R/tools/ktrace.py $OUT/prof 5 > $OUT/ktrace.txt
grep -E "blur|march" $OUT/kstats.txt
head -12 $OUT/ktrace.txt
tail -2 $OUT/prof.log
rm -rf $OUT/prof
