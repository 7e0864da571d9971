#!/bin/bash
# the first half of the previous panel as a riding tile instead of on the panel chain: parity of the factor chain, then timing
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_lml_batch_gpu.py tests/test_lml_small_gpu.py tests/test_host_mirror_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert|FAILED" | tail -8
timeout 600 python3 tools/ab_chol_overlap.py 256 1024 2048 4096 6144 7168 8192 2>&1 | tail -16
