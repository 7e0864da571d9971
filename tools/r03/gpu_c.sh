#!/bin/bash
O=gpurun_out/r03_c; mkdir -p $O
timeout 900 python -m pytest tests/test_predict_server_gpu.py -x -q -s > $O/pytest_server.log 2>&1; echo "server tests rc=$?"; grep -E "us per call|us \(device|generations|passed|failed|Error" $O/pytest_server.log | tail
timeout 600 python tools/latency_serve.py > $O/latency_serve.md 2>&1; echo "latency rc=$?"; cat $O/latency_serve.md
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -3 $O/pytest_all.log
