#!/bin/bash
# configs[4] farm on one GPU (bench.py --workload farm): the batched chain above 4096 rows in the throughput schedule, arena limit and stream groups
mkdir -p gpurun_out/r06
{
for cfg in "default::1" "lml_batch=8192,lml_streams=3::1" "lml_batch=8192,lml_streams=3,lml_batch_mb=98304::1" "lml_batch=8192,lml_streams=4,lml_batch_mb=98304::1" \
           "lml_batch=8192,lml_streams=2,lml_batch_mb=98304::1" "lml_batch=8192,lml_streams=2,lml_batch_mb=98304::2" "lml_batch=8192,lml_streams=3,lml_batch_mb=98304,tp_block=1024,tp_tail=2048::1"; do
  opts=${cfg%%::*}; groups=${cfg##*::}
  [ "$opts" = "default" ] && opts=""
  echo "GPRY_HIP_OPTIONS=$opts GPRY_HIP_FIT_TP_GROUPS=$groups"
  GPRY_HIP_OPTIONS="$opts" GPRY_HIP_FIT_TP_GROUPS=$groups timeout 900 python3 bench.py --workload farm --steps 1 --warmup 0 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('restarts/s', r['value'], 'ms_per_step', r['ms_per_step'], 'evals', r['farm']['lml_grad_evals_per_step_per_rank'], 'ms_per_eval_wall', r['roofline']['ms_per_eval_wall'], 'frac', r['roofline']['frac'])"
done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/farm2.log
