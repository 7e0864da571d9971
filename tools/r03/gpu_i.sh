#!/bin/bash
O=gpurun_out/r03_i; mkdir -p $O
timeout 900 python -m pytest tests/test_lml_small_gpu.py tests/test_host_mirror_gpu.py -x -q 2>&1 | tail -3
timeout 300 python tools/latency_lml.py 2>&1 | head -4
timeout 600 python tests/tools/stress_concurrent_fit.py 2>&1 | tail -5
