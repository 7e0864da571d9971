#!/bin/bash
mkdir -p gpurun_out/r06
{
timeout 1500 python3 -m pytest tests/test_host_mirror_gpu.py -x -q -m gpu -k "throughput_schedule or side_by_side or f6" 2>&1 | tail -5
for a in "8192 20 4 2" "8192 20 4 1" "8192 20 8 2" "6144 16 8 2" "5120 16 8 2"; do
  timeout 600 python3 tools/r06/check_tp.py $a
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp9.log
