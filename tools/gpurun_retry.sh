#!/bin/bash
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'   -- retries while the pod's GPU slots are busy (exit code 3)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
