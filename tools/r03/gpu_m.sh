#!/bin/bash
# alternating k walk: A/B timing + FETCH_SIZE per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_m; mkdir -p $O
timeout 600 python3 tools/ab_altwalk.py > $O/ab_altwalk.log 2>&1
export TRAFFIC_TAG=r03
GPRY_SWEEP_ALTWALK=0 bash tools/pmc_sweep_traffic.sh "3:0" > $O/traffic.log 2>&1
GPRY_SWEEP_ALTWALK=1 bash tools/pmc_sweep_traffic.sh "3:0" >> $O/traffic.log 2>&1
cat $O/ab_altwalk.log $O/traffic.log
