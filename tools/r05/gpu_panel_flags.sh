#!/bin/bash
# experiments on the panel step by flag (GPRY_PANEL_FLAGS): potrf times and the chain / tail summary of the stamped build
mkdir -p gpurun_out/r05
for f in 0 1 2 3; do
  echo "== flags $f"
  GPRY_PANEL_FLAGS=$f timeout 200 python3 tools/r04/time_potrf.py 1024 2048 4096 8192 2>&1 | tail -1
  GPRY_PANEL_FLAGS=$f timeout 100 python3 tools/r05/panel_stamps.py 1024 2>&1 | grep -A11 "first off" | grep -E "wave 0|wave 1:|wave 3|wave 4|start ->"
done 2>&1 | tee gpurun_out/r05/panel_flags.log
