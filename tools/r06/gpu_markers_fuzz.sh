#!/bin/bash
mkdir -p gpurun_out/r06
bash tools/r06/gpu_markers.sh 2>&1 | tail -40
cd $GRAFT_REPO_ROOT
{
  echo "# tests/tools/fuzz_parity.py 300 cases x 2 seeds (with the throughput schedule of the batched objective), fuzz_mirror.py 60 sequences, fuzz_gates.py 40 models (build of round 6: one-chain factor arithmetic, hybrid panel)"
  for seed in 61 62; do timeout 1500 python3 tests/tools/fuzz_parity.py 300 $seed 2>&1 | tail -4; done
  timeout 1200 python3 tests/tools/fuzz_mirror.py 60 63 2>&1 | tail -3
  timeout 900 python3 tests/tools/fuzz_gates.py 40 64 2>&1 | tail -3
} > gpurun_out/r06/fuzz.log 2>&1
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/r06/fuzz.log
