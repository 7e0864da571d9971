#!/bin/bash
mkdir -p gpurun_out/r06
{
for a in "4096 16 32 2" "4096 16 40 2" "4096 16 42 2" "4096 16 48 2" "4096 16 42 3" "4096 16 18 2" "4096 16 18 1" "4096 16 16 1" "4096 16 21 1" "4096 16 24 1"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp7.log
bash tools/r06/pmc_factor.sh 2>&1 | tail -40 | tee gpurun_out/r06/pmc_summary.txt
