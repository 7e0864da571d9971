#!/bin/bash
# round 4, final build: GPU suite, bench lines (default, config1) + the same under rocprofv3, fit timings
mkdir -p gpurun_out/r04
bash tools/r04/gpu_bench_profile.sh > gpurun_out/r04/gpu_bench_profile.out 2>&1
timeout 1500 python3 tools/r04/time_lml_batch.py > gpurun_out/r04/time_lml_batch.log 2>&1
timeout 1500 python3 tools/r04/time_fit_crossover.py 200 400 800 1024 1600 2048 3072 4096 2> /dev/null | grep "^N=" > gpurun_out/r04/fit_groups.log
tail -4 gpurun_out/r04/all_gpu_tests.log; tail -9 gpurun_out/r04/fit_groups.log; tail -7 gpurun_out/r04/time_lml_batch.log
timeout 600 python3 tools/r04/time_potrf.py 512 1024 2048 4096 5120 6144 7168 8192 2>&1 | tail -1 > gpurun_out/r04/time_potrf.log
timeout 900 python3 bench.py --workload farm --steps 1 --warmup 0 > gpurun_out/r04/farm.json 2> gpurun_out/r04/farm.err
cat gpurun_out/r04/time_potrf.log; head -c 300 gpurun_out/r04/farm.json
