#!/bin/bash
# per-kernel time of the batched objective (rocprofv3 --kernel-trace --stats), two shapes
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest $R/tests/test_lml_batch_gpu.py -x -q -m gpu 2>&1 | tail -40 > $R/gpurun_out/r04/batch_tests.log
for shape in "1024 8 32" "1600 8 32" "400 6 22"; do
  set -- $shape
  rm -rf /tmp/prof_$1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o p -- python3 $R/tools/r04/prof_lml_batch.py $1 $2 $3 10 > $R/gpurun_out/r04/prof_batch_$1.log 2>&1
  f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/r04/prof_batch_$1_kernel_stats.csv
done
cat $R/gpurun_out/r04/batch_tests.log
