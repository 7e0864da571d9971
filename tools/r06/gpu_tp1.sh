#!/bin/bash
# first look at the throughput schedule: parity tests that touch the factor chain, then correctness + timing of check_tp.py
mkdir -p gpurun_out/r06
{
timeout 1200 python3 -m pytest tests/test_hip_parity.py tests/test_lml_batch_gpu.py -x -q -m gpu 2>&1 | tail -8
for a in "1024 8 32 1" "1024 8 32 2" "1024 8 32 4" "1024 8 8 2" "2048 16 32 1" "2048 16 32 2" "2048 16 32 4" "4096 16 16 1" "4096 16 16 2" "4096 16 16 4" "1600 8 32 2" "400 6 32 2"; do
  timeout 300 python3 tools/r06/check_tp.py $a
done
} 2>&1 | tee gpurun_out/r06/tp1.log
