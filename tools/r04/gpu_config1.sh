#!/bin/bash
# bench line of BASELINE configs[1] and the same command under rocprofv3
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --workload config1 --steps 20 --warmup 5 > $R/gpurun_out/r04/bench_config1.json 2> $R/gpurun_out/r04/bench_config1.err
rm -rf /tmp/prof_c1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c1 -o p -- python3 $R/bench.py --workload config1 --steps 20 --warmup 5 --cpu-baseline off > $R/gpurun_out/r04/bench_config1_under_rocprof.json 2> /dev/null
cp $(find /tmp/prof_c1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r04/bench_config1_kernel_stats.csv
cat $R/gpurun_out/r04/bench_config1.json; tail -3 $R/gpurun_out/r04/bench_config1.err
