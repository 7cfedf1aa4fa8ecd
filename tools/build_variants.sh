#!/bin/bash
# Here (no GPU): build keypoint.hip experiment variants into build_variants/vN.so (travels to the GPU box).
# usage: tools/build_variants.sh "<flags1>" "<flags2>" ...
R=/root/repo
mkdir -p $R/build_variants; rm -f $R/build_variants/*.so $R/build_variants/flags.txt
cd $R/popsift_amd/csrc && make -s || exit 1
i=0
for V in "$@"; do
  i=$((i+1))
  OBJS=""
  for f in $(sed -n 's/^SRCS *= *//p' Makefile); do
    b=${f%.hip}
    X=$(sed -n "s/^FLAGS_$b *= *//p" Makefile)   # the Makefile's per-file flags
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $X $V -c $f -o /tmp/var_$b.o || exit 1
    OBJS="$OBJS /tmp/var_$b.o"
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/v$i.so $OBJS || exit 1
  echo "v$i: $V" >> $R/build_variants/flags.txt
done
cat $R/build_variants/flags.txt
