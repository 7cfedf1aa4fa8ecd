#!/bin/bash
# GPU box: bench.py under different runtime knobs (hardware queues, persistent grid sizes)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { echo "== $*"; env "$@" python bench.py --steps 6 --warmup 2 --quick 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value %.1f Mpix/s  ms/step %.3f' % (d['value'], d['ms_per_step']))"; }
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=3
run GPU_MAX_HW_QUEUES=4
run A=1
