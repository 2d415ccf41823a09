#!/bin/bash
# local helper (build container only): gpurun with retries while every GPU slot of the pod is busy (exit code 3 = nothing charged)
# usage: tools/gpurun_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
