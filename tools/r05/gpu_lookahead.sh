#!/bin/bash
# round 5: look-ahead over the outer blocks of the Cholesky (Np >= 5120): debug run, bit-identity against the comparator, stage times
mkdir -p gpurun_out/r05
timeout 60 python3 tools/r05/debug_panel.py 100 1000 2>&1 | grep "N=" | tee gpurun_out/r05/debug_panel.log
grep -q "N=1000: factorize -> 0" gpurun_out/r05/debug_panel.log || { echo "debug run failed: stopping"; exit 1; }
timeout 120 python3 tools/r05/debug_panel.py 5120 8192 2>&1 | grep "N=" | tee gpurun_out/r05/debug_panel_la.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "riding or full_size_properties or pipelined_factor" 2>&1 | tail -3 | tee gpurun_out/r05/la_tests.log
timeout 300 python3 tools/r04/time_potrf.py 4096 5120 6144 7168 8192 2>&1 | tail -1 | tee gpurun_out/r05/la_time_potrf.log
GPRY_SET=chol_lookahead=0 timeout 300 python3 tools/r04/time_potrf.py 5120 6144 7168 8192 2>&1 | tail -1 | tee gpurun_out/r05/la_time_potrf_off.log
