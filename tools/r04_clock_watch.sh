#!/bin/bash
# GPU box: shader clock and package power (rocm-smi) sampled while a command runs.   tools/r04_clock_watch.sh <label> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}
label=$1; shift
cd $R
"$@" > /dev/null 2>&1 &
pid=$!
n=0
while kill -0 $pid 2>/dev/null; do
  out=$(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ' ')
  echo "$label t=$n $out"
  n=$((n+1))
  sleep 0.2
done
