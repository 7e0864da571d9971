#!/bin/bash
O=gpurun_out/r03_b; mkdir -p $O
timeout 900 python -m pytest tests/test_predict_server_gpu.py -x -q -s > $O/pytest_server.log 2>&1; echo "server tests rc=$?"; tail -15 $O/pytest_server.log
timeout 600 python tools/latency_serve.py > $O/latency_serve.md 2>&1; echo "latency rc=$?"; cat $O/latency_serve.md
