#!/bin/bash
# round 5: stream-K launches of V = L^-1 on the side stream with a bounded number of workgroups (option factor_side_wgs)
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/side_wgs.log
GPRY_HIP_OPTIONS=factor_side_wgs=128 timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pipelined" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee gpurun_out/r05/side_wgs_tests.log
for rep in 1 2; do
for w in 0 48 64 96 128 192 256 384; do
  echo "side_wgs=$w: $(GPRY_HIP_OPTIONS=factor_side_wgs=$w timeout 300 python3 tools/r04/ab_pipeline_now.py 4096 5120 2>&1 | grep '^N=' | sed 's/pipeline=0: [^|]*| //' | tr '\n' ';')" | tee -a gpurun_out/r05/side_wgs.log
done
done
