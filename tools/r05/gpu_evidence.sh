#!/bin/bash
# round 5: evidence of the build -- GPU suite, bench lines (default / config1 / farm) and the default + config1 commands under
# rocprofv3 (kernel stats), PMC passes (matrix-pipe busy of the factor chain; HBM-side traffic of the sweep), stage timers
mkdir -p gpurun_out/r05
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 2400 python -m pytest $R/tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $R/gpurun_out/r05/all_gpu_tests.log
cat $R/gpurun_out/r05/all_gpu_tests.log
timeout 1200 python3 $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/r05/bench_final.json 2> $R/gpurun_out/r05/bench_final.err
rm -rf /tmp/prof_b
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o p -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-baseline off --extras off > $R/gpurun_out/r05/bench_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05/bench_kernel_stats.csv
timeout 600 python3 $R/bench.py --workload config1 --steps 20 --warmup 5 > $R/gpurun_out/r05/bench_config1.json 2> $R/gpurun_out/r05/bench_config1.err
rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o p -- python3 $R/bench.py --workload config1 --steps 20 --warmup 5 --cpu-baseline off > $R/gpurun_out/r05/bench_config1_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r05/bench_config1_kernel_stats.csv
timeout 900 python3 $R/bench.py --workload farm --steps 1 --warmup 0 > $R/gpurun_out/r05/farm.json 2> $R/gpurun_out/r05/farm.err
cd $R
bash tools/r05/pmc_traffic.sh > /dev/null 2>&1
bash tools/r05/pmc_factor.sh > /dev/null 2>&1
timeout 600 python3 tools/r04/time_potrf.py 512 1024 2048 3072 4096 5120 6144 7168 8192 2>&1 | tail -1 > gpurun_out/r05/time_potrf.log
timeout 900 python3 tools/r04/time_lml_batch.py > gpurun_out/r05/time_lml_batch.log 2>&1
timeout 900 python3 tools/r04/time_fit_crossover.py 200 400 800 1024 1600 2048 3072 4096 2> /dev/null | grep "^N=" > gpurun_out/r05/fit_groups.log
timeout 300 python3 tools/r04/ab_pipeline_now.py 4096 8192 2>&1 | grep "^N=" > gpurun_out/r05/ab_pipeline.log
cat gpurun_out/r05/time_potrf.log gpurun_out/r05/pmc_traffic.txt; tail -12 gpurun_out/r05/time_lml_batch.log; cat gpurun_out/r05/fit_groups.log gpurun_out/r05/ab_pipeline.log; head -c 400 gpurun_out/r05/farm.json
