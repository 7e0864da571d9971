#!/bin/bash
O=gpurun_out/r03_d; mkdir -p $O
timeout 900 python -m pytest tests/test_lml_small_gpu.py -x -q > $O/pytest_small.log 2>&1; echo "small tests rc=$?"; tail -25 $O/pytest_small.log
for s in 1 0; do echo "lml_small=$s"; GPRY_HIP_OPTIONS=lml_small=$s timeout 300 python tools/latency_lml.py 2>&1 | head -4; done
