#!/bin/bash
# Here (no GPU): build keypoint.hip experiment variants into build_variants/vN.so (travels to the GPU box).
# usage: tools/build_variants.sh "<flags1>" "<flags2>" ...
R=/root/repo
mkdir -p $R/build_variants; rm -f $R/build_variants/*.so $R/build_variants/flags.txt
cd $R/popsift_amd/csrc && make -s || exit 1
i=0
for V in "$@"; do
  i=$((i+1))
  for f in ctx pyramid extrema keypoint; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $V -c $f.hip -o /tmp/var_$f.o || exit 1
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/v$i.so /tmp/var_ctx.o /tmp/var_pyramid.o /tmp/var_extrema.o /tmp/var_keypoint.o || exit 1
  echo "v$i: $V" >> $R/build_variants/flags.txt
done
cat $R/build_variants/flags.txt
