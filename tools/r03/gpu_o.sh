#!/bin/bash
# full GPU suite on the build with the alternating k walk as default, then the bench + rocprof of gpu_final.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_o; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_all.log
bash tools/r03/gpu_final.sh
cat gpurun_out/r03_final/bench.json
