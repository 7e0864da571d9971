#!/bin/bash
mkdir -p gpurun_out/r06
{
timeout 900 python3 tools/r06/ab_sweep_overlap.py 4096 16 1000000
timeout 900 python3 tools/r06/ab_sweep_overlap.py 1024 8 400000
for o in 1 0; do echo "sweep_overlap=$o"; GPRY_HIP_OPTIONS="sweep_overlap=$o" timeout 900 python3 bench.py --steps 8 --warmup 3 --extras off --cpu-baseline off 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['cycle']
print('ms_per_step', r['ms_per_step'], 'resident', c['resident_pool_ms_per_step'], 'roofline', r['roofline']['frac'], 'panel', c.get('panel_form'), 'cross_build', c['stage_ms_per_step']['cross_build'], 'sweep_gemm', c['stage_ms_per_step']['sweep_gemm'], 'acq', c['acquisition_ms'])"; done
GPRY_HIP_OPTIONS="sweep_overlap=1" timeout 1500 python3 -m pytest tests/test_hip_parity.py tests/test_group_gpu.py tests/test_host_mirror_gpu.py -x -q -m gpu -k "sweep or config2 or f7 or group or multi_add or upload or fitted" 2>&1 | tail -5
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/ov.log
