#!/bin/bash
# randomised parity sweeps of the final build, longer than the suite runs them (profiles/r04_fuzz.log)
mkdir -p gpurun_out/r05
{
  echo "# tests/tools/fuzz_parity.py 300 cases x 2 seeds, fuzz_mirror.py 60 sequences, fuzz_gates.py 40 models (final build of round 5)"
  for seed in 51 52; do timeout 1500 python3 tests/tools/fuzz_parity.py 300 $seed 2>&1 | tail -4; done
  timeout 1200 python3 tests/tools/fuzz_mirror.py 60 53 2>&1 | tail -3
  timeout 900 python3 tests/tools/fuzz_gates.py 40 54 2>&1 | tail -3
} > gpurun_out/r05/fuzz.log 2>&1
cat gpurun_out/r05/fuzz.log
