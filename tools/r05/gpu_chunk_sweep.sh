#!/bin/bash
# the cycle by size of the sweep chunk (panel of Np x chunk doubles: 1 GiB at 32768, inside the 256-MB Infinity Cache from 8192 down)
mkdir -p gpurun_out/r05
for c in 4096 8192 16384 32768 65536; do
  GPRY_HIP_OPTIONS="sweep_chunk=$c" timeout 300 python3 bench.py --steps 5 --warmup 2 --extras off --cpu-baseline off 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); s=r['cycle']['stage_ms_per_step']
print('chunk $c: cycle %.2f ms; cross_build %.2f, sweep_gemm %.2f (%.3f of peak), finish %.2f' % (r['ms_per_step'], s['cross_build'], s['sweep_gemm'], r['roofline']['frac'], s['sweep_finish']))"
done | tee gpurun_out/r05/chunk_sweep.log
