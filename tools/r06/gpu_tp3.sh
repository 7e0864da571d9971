#!/bin/bash
# tile-major dispatch order + minimal clearing: batched timings again; then full fits under both schedules
mkdir -p gpurun_out/r06
{
for a in "1024 8 32 2" "2048 16 32 2" "4096 16 16 2" "4096 16 16 1" "4096 16 16 3" "2048 16 32 3" "1600 8 32 2" "4096 16 8 2" "2048 16 16 2"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
export GPRY_HIP_FIT_SCHEDULE=latency; timeout 600 python3 tools/r06/time_fit.py 4096 16 2
export GPRY_HIP_FIT_SCHEDULE=throughput
for g in 1 2 3; do for st in 2 3; do
  GPRY_HIP_FIT_TP_GROUPS=$g GPRY_TP_STREAMS=$st timeout 600 python3 tools/r06/time_fit.py 4096 16 2
done; done
export GPRY_HIP_FIT_SCHEDULE=latency; timeout 600 python3 tools/r06/time_fit.py 2048 16 2; timeout 600 python3 tools/r06/time_fit.py 1024 8 3
export GPRY_HIP_FIT_SCHEDULE=throughput
for g in 1 2; do
  GPRY_HIP_FIT_TP_GROUPS=$g timeout 600 python3 tools/r06/time_fit.py 2048 16 2
  GPRY_HIP_FIT_TP_GROUPS=$g timeout 600 python3 tools/r06/time_fit.py 1024 8 3
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp3.log
