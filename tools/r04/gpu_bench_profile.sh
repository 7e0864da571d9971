#!/bin/bash
# round 4: GPU suite, bench line of the build, the same command under rocprofv3 (kernel stats), config1 line + profile
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $R/gpurun_out/r04/all_gpu_tests.log
timeout 1200 python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r04/bench_final.json 2> $R/gpurun_out/r04/bench_final.err
rm -rf /tmp/prof_b
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o p -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-baseline off --extras off > $R/gpurun_out/r04/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04/bench_kernel_stats.csv
timeout 600 python3 $R/bench.py --workload config1 --steps 20 --warmup 5 > $R/gpurun_out/r04/bench_config1.json 2> $R/gpurun_out/r04/bench_config1.err
rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o p -- python3 $R/bench.py --workload config1 --steps 20 --warmup 5 --cpu-baseline off > $R/gpurun_out/r04/bench_config1_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04/bench_config1_kernel_stats.csv
tail -3 $R/gpurun_out/r04/all_gpu_tests.log; head -c 600 $R/gpurun_out/r04/bench_final.json; echo; head -c 400 $R/gpurun_out/r04/bench_config1.json
