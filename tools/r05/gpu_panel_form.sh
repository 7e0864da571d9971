#!/bin/bash
mkdir -p gpurun_out/r05
GPRY_HIP_DEBUG_PANEL=1 timeout 600 python3 bench.py --steps 3 --warmup 1 --extras off --cpu-baseline off 2> gpurun_out/r05/panel_form.err | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['cycle']['stage_ms_per_step']['cross_build'])"
grep "panel form" gpurun_out/r05/panel_form.err | sort | uniq -c | head -5
