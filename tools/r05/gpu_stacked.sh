#!/bin/bash
# round 5: the inverse factor as extra rows of the panel chain (option chol_stacked) -- tests, batched objective and fits with / without
mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests/test_hip_parity.py tests/test_lml_batch_gpu.py tests/test_host_mirror_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED" | tail -30 | tee gpurun_out/r05/stacked_tests.log
rm -f gpurun_out/r05/stacked_ab.log
for mode in 3584 0; do
  for s in "1024 8 2" "1024 8 8" "1024 8 32" "1024 8 64" "400 6 22" "256 4 8" "1600 8 32" "2048 16 32" "2048 16 4"; do
    echo "chol_stacked=$mode: $(GPRY_HIP_OPTIONS=chol_stacked=$mode timeout 200 python3 tools/r04/prof_lml_batch.py $s 20 | tail -1)" | tee -a gpurun_out/r05/stacked_ab.log
  done
  GPRY_HIP_OPTIONS=chol_stacked=$mode timeout 900 python3 tools/r04/time_fit_crossover.py 200 400 1024 1600 2048 2> /dev/null | grep "^N=" | sed "s/^/chol_stacked=$mode: /" | tee -a gpurun_out/r05/stacked_ab.log
done
