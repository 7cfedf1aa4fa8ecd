#!/bin/bash
# GPU box: per-kernel average durations inside the timed loop of bench.py --quick (3 contexts x 8 images per launch).
#   tools/r04_bench_kstats.sh <tag> "<debug switches or ->" [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/bk_${1:-x}
DBG=$2; shift; shift
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ "$DBG" = "-" ] && DBG=""
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --quick --steps 6 --warmup 2 ${DBG:+--debug $DBG} "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 $R/tools/kstats.py $OUT/prof > $OUT/kstats.txt
tail -1 $OUT/bench.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms_per_step", d["ms_per_step"])'
head -24 $OUT/kstats.txt
rm -rf $OUT/prof
