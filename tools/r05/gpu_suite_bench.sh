#!/bin/bash
# round 5: whole GPU suite, default bench line, potrf stage times
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/all_gpu_tests.log
cat gpurun_out/r05/all_gpu_tests.log
timeout 600 python3 bench.py --steps 10 --warmup 3 > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
head -c 1500 gpurun_out/r05/bench_default.json
timeout 300 python3 tools/r04/time_potrf.py 512 1024 2048 3072 4096 6144 8192 2>&1 | tail -1 | tee gpurun_out/r05/time_potrf.log
