#!/bin/bash
mkdir -p gpurun_out/r05
for cfg in "256 0" "0 0" "0 1" "512 0" "384 0" "128 0"; do
  set -- $cfg
  if [ "$2" = 1 ]; then export GPRY_LA_PRIO=1; else unset GPRY_LA_PRIO; fi
  echo "walk=$1 prio=$2: $(GPRY_LA_WALK=$1 timeout 300 python3 tools/r04/time_potrf.py 5120 6144 8192 2>&1 | tail -1)" | tee -a gpurun_out/r05/la_sweep.log
done
