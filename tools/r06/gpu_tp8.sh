#!/bin/bash
mkdir -p gpurun_out/r06
{
timeout 1500 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "fitted_model_of_the_bench or panel_form or config2 or f4_ or f7 or group" 2>&1 | tail -8
for h in 1 0; do echo "cross_hybrid=$h"; GPRY_HIP_OPTIONS="cross_hybrid=$h" timeout 900 python3 bench.py --steps 8 --warmup 3 --extras off --cpu-baseline off 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['cycle']
print('ms_per_step', r['ms_per_step'], 'panel', c.get('panel_form'), c.get('panel_error_estimate'), c.get('panel_error_variance'), 'cross_build', c['stage_ms_per_step']['cross_build'], 'sweep_gemm', c['stage_ms_per_step']['sweep_gemm'])"; done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tp8.log
