#!/bin/bash
mkdir -p gpurun_out/r06
{
for a in "4096 16 1 2" "4096 16 2 2" "4096 16 3 2" "4096 16 4 2" "4096 16 6 2" "4096 16 10 2" "4096 16 16 2" "2048 16 32 2" "1024 8 32 2"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
GPRY_HIP_FIT_SCHEDULE=throughput timeout 600 python3 tools/r06/time_fit.py 4096 16 2
GPRY_HIP_FIT_SCHEDULE=latency timeout 600 python3 tools/r06/time_fit.py 4096 16 2
GPRY_HIP_FIT_SCHEDULE=throughput timeout 600 python3 tools/r06/time_fit.py 4096 16 2
GPRY_HIP_FIT_SCHEDULE=throughput GPRY_HIP_FIT_TP_GROUPS=2 timeout 600 python3 tools/r06/time_fit.py 4096 16 2
# the sweep with and without the trimmed diagonal block (GPRY_SWEEP_TRIM: a switch of the experimental build of profiles/r06_sweep.md (a), not in the tree any more): one resident pool of 1e6 at N = 4096
for t in 1 0 1 0; do echo trim=$t; GPRY_SWEEP_TRIM=$t timeout 600 python3 tools/prof_sweep.py 4096 16 1048576 2>&1 | tail -4; done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp6.log
