#!/bin/bash
# round 5, iteration on the panel step: debug run, factor tests, potrf stage times, section stamps
mkdir -p gpurun_out/r05
timeout 60 python3 tools/r05/debug_panel.py 100 200 1000 2>&1 | grep "N=" | tee gpurun_out/r05/debug_panel.log
grep -q "N=1000: factorize -> 0" gpurun_out/r05/debug_panel.log || { echo "debug run failed: stopping"; exit 1; }
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "f2_factor or non_positive or full_size_properties or ragged_block or riding or pipelined_factor or bordered" 2>&1 | tail -3 | tee gpurun_out/r05/factor_tests.log
timeout 300 python3 tools/r04/time_potrf.py 512 1024 2048 3072 4096 6144 8192 2>&1 | tail -1 | tee gpurun_out/r05/time_potrf.log
for N in 1024 4096; do timeout 120 python3 tools/r05/panel_stamps.py $N 2>&1 | tee gpurun_out/r05/stamps_$N.log; done
