#!/bin/bash
# Runs on the GPU box (via gpurun): what binds the TIMED LOOP of bench.py (dense images only, `--quick`).
#   1. rocprofv3 --kernel-trace of `python3 bench.py --quick` -> tools/mix_timeline.py (who runs beside whom)
#   2. one --pmc pass of the same command (issue utilisation: vector instructions of every kernel of the loop)
# usage: mix_evidence.sh <tag> [bench args]   ->  gpurun_out/mix_<tag>/{timeline.txt,kernel_stats.txt,pmc.txt,bench_*.json}
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
shift
ARGS=${@:---steps 6 --warmup 2}
RAW=/tmp/mix_raw_$TAG
OUT=$R/gpurun_out/mix_$TAG
rm -rf $RAW $OUT
mkdir -p $RAW $OUT
cd /tmp && export TMPDIR=/tmp
# the untraced figure first (the tracer costs a few per cent)
(cd $R && timeout -k 10 300 python3 bench.py --quick $ARGS > $OUT/bench_plain.json 2> $OUT/bench_plain.err) || exit 1
cat $OUT/bench_plain.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -- python3 $R/bench.py --quick $ARGS > $OUT/bench_traced.json 2> $RAW/trace.err || exit 1
cat $OUT/bench_traced.json
python3 $R/tools/mix_timeline.py $RAW/trace > $OUT/timeline.txt 2>&1
python3 $R/tools/kstats.py $RAW/trace > $OUT/kernel_stats.txt 2>&1
head -30 $OUT/timeline.txt
[ -n "$NOPMC" ] && exit 0
timeout -k 10 500 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $RAW/pmc -- python3 $R/bench.py --quick --steps 2 --warmup 1 > $OUT/bench_pmc.json 2> $RAW/pmc.err
echo "pmc rc=$?"
python3 $R/tools/pmc.py $RAW/pmc > $OUT/pmc.txt 2>&1
cat $OUT/pmc.txt
