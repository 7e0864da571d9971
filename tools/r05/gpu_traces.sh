#!/bin/bash
# round 5: lml_traces_kernel with the coordinate loop outermost -- same bits (gradient goldens, batch tests), stage time
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_lml_batch_gpu.py -x -q -m gpu -k "lml or grad or batch or objective" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r05/traces_tests.log
timeout 300 python3 tools/r05/time_traces.py 1024 8 2048 16 4096 16 4096 4 8192 20 4096 32 2>&1 | tail -1 | tee gpurun_out/r05/traces_time.log
