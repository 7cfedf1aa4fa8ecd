#!/bin/bash
# GPU box: rows per segment of the march kernels (debug switch 9) at the bench's launch size: summed k_blur_march time per image
# (one context, rocprofv3) and bench.py --quick dense / keypoint-sparse (three contexts).   tools/r04_seg_sweep.sh <rows> ...  (0 = the library's choice)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for seg in "$@"; do
  dbg=""; [ "$seg" != "0" ] && dbg="9:$seg"
  tools/r04_bench_kstats.sh seg$seg "${dbg:--}" --contexts 1 > /dev/null 2>&1
  m=$(python3 - <<P
import re
tot=0; calls=0
for l in open("$R/gpurun_out/bk_seg$seg/kstats.txt"):
    if 'k_blur_march' in l:
        mm=re.search(r'calls\s+(\d+) tot\s+([\d.]+) us', l); tot+=float(mm.group(2))
# 32 batches of 16 images in the profiled run (6 steps + 2 warm-up x 64 images)
print("%.1f" % (tot/ (8*64)))
P
)
  d=$(timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 ${dbg:+--debug $dbg} 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')
  s=$(timeout -k 10 300 python3 bench.py --quick --steps 10 --warmup 2 ${dbg:+--debug $dbg} --threshold 0.17 2>/dev/null | tail -1 | python3 -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])')
  echo "seg $seg: march us per image $m   dense $d   sparse $s"
done
