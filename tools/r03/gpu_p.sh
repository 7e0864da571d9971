#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_p; mkdir -p $O
timeout 600 python3 tools/ab_kernel_build.py > $O/ab_kernel_build.md 2>&1; cat $O/ab_kernel_build.md
