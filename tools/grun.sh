#!/bin/bash
# Here (build container): gpurun with retries while no GPU slot is free (exit code 3: nothing charged).
#   tools/grun.sh <timeout> '<command>' <logfile>
T=$1; CMD=$2; LOG=$3
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$CMD" > $LOG 2>&1
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
