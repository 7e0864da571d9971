#!/bin/bash
mkdir -p gpurun_out/r06
{
for a in "4096 16 16 2 512 1024 1" "4096 16 16 2 512 1024 0" "4096 16 16 2 512 1024 1" "4096 16 16 2 512 1024 0" "4096 16 16 2 256 512 1" "4096 16 16 2 1024 1024 1" "4096 16 16 2 512 512 1" "2048 16 32 2 512 1024 1" "2048 16 32 2 512 1024 0" "2048 16 32 2 256 512 1" "2048 16 32 2 512 512 1" "4096 16 16 1 512 1024 1" "4096 16 16 1 512 1024 0"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
timeout 1500 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "riding or underneath_the_contraction" 2>&1 | grep -E "passed|failed|error" | tail -3
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/left.log
