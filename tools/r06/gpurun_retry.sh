#!/bin/bash
# gpurun with retries while no slot is free (exit code 3): gpurun_retry.sh <timeout> <log> <command>
t=$1; log=$2; shift 2
for i in $(seq 1 30); do
  gpurun --timeout $t -- "$@" > $log 2>&1
  rc=$?
  if ! grep -q "status=transient" $log; then exit $rc; fi
  sleep 120
done
