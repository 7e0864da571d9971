#!/bin/bash
O=gpurun_out/r03_g; mkdir -p $O
timeout 600 python -m pytest tests/test_host_mirror_gpu.py -x -q -k "f9 or f6 or f7" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_lml_small_gpu.py tests/test_hip_parity.py -x -q -k "small or f1_ or f2_ or f8 or bordered" 2>&1 | tail -3
timeout 300 python tools/ab_kernel_build.py 2>&1 | head -9
