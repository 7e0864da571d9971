#!/bin/bash
# round 5: where the stacked inverse factor stops paying -- fits and batched calls at N = 2048 / 3072 with chol_stacked = 0 / 2048 / 3584
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/stacked_limit.log
for mode in 0 3584; do
  for s in "3072 8 4" "3072 8 16" "2560 8 8" "2560 8 2"; do
    echo "chol_stacked=$mode: $(GPRY_HIP_OPTIONS=chol_stacked=$mode timeout 200 python3 tools/r04/prof_lml_batch.py $s 10 | tail -1)" | tee -a gpurun_out/r05/stacked_limit.log
  done
  GPRY_HIP_OPTIONS=chol_stacked=$mode timeout 900 python3 tools/r04/time_fit_crossover.py 2048 3072 2> /dev/null | grep "^N=" | sed "s/^/chol_stacked=$mode: /" | tee -a gpurun_out/r05/stacked_limit.log
done
