#!/bin/bash
# round 3: bench line of the final build, the same command under rocprofv3, small-N tables
O=$GRAFT_REPO_ROOT/gpurun_out/r03_final; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python3 tools/latency_small_n.py > $O/latency_small_n.md 2>&1; echo "latency rc=$?"
timeout 300 python3 tools/lml_small_sections.py > $O/lml_small_sections.log 2>&1
timeout 300 python3 tools/latency_lml.py > $O/latency_lml.log 2>&1
timeout 300 python3 tools/latency_serve.py > $O/latency_serve.md 2>&1
timeout 300 python3 tools/latency_gates.py > $O/latency_gates.log 2>&1
export TMPDIR=/tmp
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-baseline off --extras off > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
rm -rf $O/prof
ls -la $O
