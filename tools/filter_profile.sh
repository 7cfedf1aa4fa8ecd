#!/bin/bash
# GPU box: kernel times of the grid-filter stage on the config-2 image (77 k extrema -> 20 k)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/filter_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PROF_KW='{"filter_max_extrema": 20000, "filter_sorting": 1, "filter_grid_size": 4}'
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/prof_run.py 5 > $OUT.log 2>&1 || exit 1
python3 $R/tools/kstats.py $OUT
