#!/bin/bash
O=gpurun_out/r03_e; mkdir -p $O
timeout 900 python -m pytest tests/test_lml_small_gpu.py -x -q > $O/pytest_small.log 2>&1; echo "small tests rc=$?"; tail -4 $O/pytest_small.log
timeout 300 python tools/lml_small_sections.py
