#!/bin/bash
# where the side-by-side fit should change from the latency to the throughput schedule (gpry_amd/gpr.py: fit_schedule -- 2560 padded rows)
mkdir -p gpurun_out/r06
{
for nd in "2304 16" "2560 16" "3072 16" "3584 16" "3072 8" "2560 4" "4096 4"; do
  for s in latency throughput; do
    GPRY_HIP_FIT_SCHEDULE=$s timeout 900 python3 tools/r06/time_fit.py $nd 2 2>&1 | grep "^N="
  done
done
} 2>&1 | tee gpurun_out/r06/fit_threshold.log
