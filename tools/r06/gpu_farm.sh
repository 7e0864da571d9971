#!/bin/bash
mkdir -p gpurun_out/r06
{
for cfg in "default::auto" "lml_batch=8192::auto" "lml_batch=8192::latency" "lml_batch=8192,lml_streams=3::auto"; do
  opts=${cfg%%::*}; sched=${cfg##*::}
  [ "$opts" = "default" ] && opts=""
  echo "GPRY_HIP_OPTIONS=$opts GPRY_HIP_FIT_SCHEDULE=$sched"
  GPRY_HIP_OPTIONS="$opts" GPRY_HIP_FIT_SCHEDULE=$sched timeout 900 python3 bench.py --workload farm --steps 1 --warmup 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('restarts/s', r['value'], 'ms_per_step', r['ms_per_step'], 'evals', r['farm']['lml_grad_evals_per_step_per_rank'], 'ms_per_eval_wall', r['roofline']['ms_per_eval_wall'], 'frac', r['roofline']['frac'])"
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/farm_ab.log
