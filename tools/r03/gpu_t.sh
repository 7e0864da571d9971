#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_t; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $O/pytest_all.log | tail -3
timeout 300 python3 tools/latency_scan_n.py 2>&1 | grep "^N=" | tee $O/latency_scan.log
