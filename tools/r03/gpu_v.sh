#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_v; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_hip_parity.py tests/test_host_mirror_gpu.py tests/test_fuzz_parity_gpu.py -x -q -k "not full_size" 2>&1 | grep -E "passed|failed|rror" | tail -3
for v in 0 1; do echo "trtri_diag128=$v"; GPRY_HIP_OPTIONS=trtri_diag128=$v timeout 300 python3 tools/latency_scan_n.py 2>&1 | grep "^N=\(256\|384\|512\|1024\)"; done | tee $O/latency_scan.log
