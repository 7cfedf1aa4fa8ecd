#!/bin/bash
# GPU box (ONE card): the multi-rank control path of bench.py with every rank on device 0 and gloo instead of RCCL --
# rank launch, sharding, barriers, the MAX / SUM reduction, the C++ leg over "all the job's GPUs" (POPSIFT_DEVICES=0,0,..).
# The pool allows at most 6 processes on the card, so 4 ranks -- a launcher and its six ranks were counted as seven -- (the driver runs the real N = 8 on an 8-GPU node).
# A rehearsal, not a scaling number.   -> gpurun_out/bench_6rank_one_card.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
N=${1:-4}
BENCH_FORCE_DEVICE=0 BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus $N --contexts 2 --launch-batch 4 --steps 3 --warmup 1 --no-cpu-baseline \
  > gpurun_out/bench_${N}rank_one_card.json 2> gpurun_out/bench_${N}rank_one_card.err
echo "rc=$?"
python3 - <<PY
import json
d = json.loads(open("gpurun_out/bench_${N}rank_one_card.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "steps", "ms_per_step")}, d["config"])
print(d.get("host_to_host_cpp_api"))
PY
