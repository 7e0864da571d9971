#!/bin/bash
# where does the remaining K*^T traffic come from: uniform-length tiles (always aligned) as the floor
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_n; mkdir -p $O
export TRAFFIC_TAG=r03
for a in 2 3; do GPRY_SWEEP_ALTWALK=$a bash tools/pmc_sweep_traffic.sh "3:0"; done > $O/traffic.log 2>&1
for tm in 2 4; do GPRY_SWEEP_ALTWALK=1 bash tools/pmc_sweep_traffic.sh "$tm:0"; done >> $O/traffic.log 2>&1
cat $O/traffic.log
