#!/bin/bash
# after a change of the Cholesky schedule: GPU suite, potrf stage times, the factor chain with / without the pipelined V = L^-1, farm line
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r04/all_gpu_tests.log
timeout 600 python3 tools/r04/time_potrf.py 512 1024 2048 4096 5120 6144 7168 8192 2>&1 | tail -1 > gpurun_out/r04/time_potrf.log
timeout 600 python3 tools/r04/ab_pipeline_now.py 4096 6144 8192 2>&1 | grep "^N=" > gpurun_out/r04/ab_pipeline.log
timeout 900 python3 bench.py --workload farm --steps 1 --warmup 0 > gpurun_out/r04/farm.json 2> gpurun_out/r04/farm.err
cat gpurun_out/r04/all_gpu_tests.log gpurun_out/r04/time_potrf.log gpurun_out/r04/ab_pipeline.log; cut -c1-600 gpurun_out/r04/farm.json
