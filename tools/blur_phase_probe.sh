#!/bin/bash
# GPU box: kernel times of the octave-0 blur launches for the build_variants/ libraries; build them first with
#   bash tools/build_variants.sh "-DV0" "-DBLUR_NO_STORE" "-DBLUR_NO_LOAD" "-DBLUR_NO_STORE -DBLUR_NO_LOAD" "-DBLUR_NO_H -DBLUR_NO_V"
# (also -DBLUR_NO_H / -DBLUR_NO_V alone or combined with the two above)
# variants are wrong by construction: only the times matter)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/blurprobe; rm -rf $OUT; mkdir -p $OUT
cp $R/popsift_amd/libpopsift_hip.so /tmp/orig.so
cd /tmp && export TMPDIR=/tmp
for so in $R/build_variants/v*.so; do
  n=$(basename $so .so); cp $so $R/popsift_amd/libpopsift_hip.so
  echo "== $(grep "^$n:" $R/build_variants/flags.txt)"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$n -- python3 $R/tools/prof_run.py 4 > $OUT/$n.log 2>&1
  python3 $R/tools/kstats.py $OUT/$n | grep -E "0, 64"
done
cp /tmp/orig.so $R/popsift_amd/libpopsift_hip.so
