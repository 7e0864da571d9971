#!/bin/bash
# roctx marker ranges, then the whole -m gpu suite
mkdir -p gpurun_out/r06
bash tools/r06/gpu_markers.sh 2>&1 | tail -45
cd $GRAFT_REPO_ROOT
timeout 3000 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -15 | tee gpurun_out/r06/pytest_gpu.log
