#!/bin/bash
# GPU box: octave-0 level launch times of one config-2 image for several segment heights of the march kernels
# (and for every build_variants/vN.so).   tools/r04_march_sweep.sh <tag> "<seg list>"
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-sweep}
SEGS=${2:-"64 96 128 192 256"}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() {
  tag=$1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p_$tag -- python3 $R/tools/prof_run.py 5 > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; return; }
  python3 $R/tools/ktrace.py $OUT/p_$tag 5 > $OUT/$tag.trace.txt
  echo "== $tag: $(grep -E '^k_blur' $OUT/$tag.trace.txt | head -9 | awk '{printf "%s ", $4}') | pyramid ends $(grep -E '^k_detect' $OUT/$tag.trace.txt | head -1 | awk '{print $2}')"
  rm -rf $OUT/p_$tag
}
for so in product $R/build_variants/v*.so; do
  if [ $so = product ]; then unset POPSIFT_HIP_LIB; n=main; else [ -f $so ] || continue; export POPSIFT_HIP_LIB=$so; n=$(basename $so .so); echo "## $(grep "^$n:" $R/build_variants/flags.txt)"; fi
  for seg in $SEGS; do
    PROF_DEBUG="8:2,9:$seg" one ${n}_seg$seg
  done
done
