#!/bin/bash
# the GPU test suite, the default bench line and the batched-objective timings of the current build
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r04/all_gpu_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
timeout 600 python tools/r04/time_lml_batch.py > gpurun_out/r04/time_lml_batch.log 2>&1
tail -5 gpurun_out/r04/all_gpu_tests.log; head -c 1500 gpurun_out/r04/bench_default.json; echo; tail -6 gpurun_out/r04/time_lml_batch.log
