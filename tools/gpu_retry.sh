#!/bin/bash
# tools/gpu_retry.sh <timeout> <command>: gpurun, retried while no box or slot is free (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
