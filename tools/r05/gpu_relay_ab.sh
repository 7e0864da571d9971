#!/bin/bash
# round 5: the relay of the batched panel step -- bit-identity, A / B, tile budget
mkdir -p gpurun_out/r05
timeout 120 python3 tools/r04/prof_lml_batch.py 256 4 4 2 2>&1 | tail -2 | tee gpurun_out/r05/relay_sanity.log
grep -q "ms per call" gpurun_out/r05/relay_sanity.log || { echo "sanity run failed: stopping"; exit 1; }
timeout 900 python -m pytest tests/test_lml_batch_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05/relay_tests.log
rm -f gpurun_out/r05/relay_ab.log
for cfg in "0 0" "128 0" "0 0" "128 0"; do
  set -- $cfg
  for shape in "1024 8 32" "1024 8 64" "1600 8 32" "2048 16 32" "400 6 22"; do
    echo "flags=$1 budget=$2: $(GPRY_PANEL_FLAGS=$1 GPRY_TILE_BUDGET=$2 timeout 300 python3 tools/r04/prof_lml_batch.py $shape 10 2>&1 | tail -1)" | tee -a gpurun_out/r05/relay_ab.log
  done
done
