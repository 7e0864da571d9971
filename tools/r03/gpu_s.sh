#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_s; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_parity.py -x -q -k "small_tile or f2_factor or pipeline_vs_oracle or ragged_block or pipelined_factor or riding" 2>&1 | tail -5
for v in 0 32; do echo "gemm_small=$v"; GPRY_HIP_OPTIONS=gemm_small=$v timeout 300 python3 tools/latency_scan_n.py 2>&1 | grep "^N=\(256\|384\|512\|1024\)"; done | tee $O/latency_scan.log
