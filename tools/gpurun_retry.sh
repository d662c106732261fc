#!/bin/bash
# submit a gpurun call, again while the pod's GPU slots are busy (exit code 3: nothing ran, nothing was charged)
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
