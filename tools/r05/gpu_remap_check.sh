#!/bin/bash
# round 5: panel workgroups of all thetas first in the dispatch order of a batched step -- bit-identity and times
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_lml_batch_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r05/remap_tests.log
rm -f gpurun_out/r05/remap_times.log
for rep in 1 2; do
  for shape in "1024 8 32" "1024 8 64" "1600 8 32" "2048 16 32" "400 6 22" "4096 16 16"; do
    echo "$(timeout 300 python3 tools/r04/prof_lml_batch.py $shape 10 2>&1 | tail -1)" | tee -a gpurun_out/r05/remap_times.log
  done
done
bash tools/r05/gpu_trace_batch.sh 0 | head -18
