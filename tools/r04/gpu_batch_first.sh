#!/bin/bash
# first GPU pass of the batched objective: its tests, the neighbouring objective tests, timing
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lml_batch_gpu.py tests/test_lml_small_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/batch_tests.log
timeout 600 python tools/r04/time_lml_batch.py > gpurun_out/r04/time_lml_batch.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/all_gpu_tests.log
cat gpurun_out/r04/batch_tests.log gpurun_out/r04/time_lml_batch.log gpurun_out/r04/all_gpu_tests.log
