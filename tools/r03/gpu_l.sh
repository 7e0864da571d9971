#!/bin/bash
timeout 1200 python3 -m pytest tests/test_hip_parity.py tests/test_fuzz_parity_gpu.py tests/test_lml_small_gpu.py -x -q -k "not full_size" 2>&1 | grep -E "passed|failed|Error" | tail -3
for v in 1 0; do echo "lml_vectors=$v"; GPRY_HIP_OPTIONS=lml_vectors=$v timeout 300 python3 tools/latency_lml.py 2>&1 | sed -n 4,7p; done
