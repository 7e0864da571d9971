#!/bin/bash
mkdir -p gpurun_out/r05; rm -f gpurun_out/r05/sk_div.log
for div in 1 4 8 16 32 512; do
  for shape in "1024 8 32" "1600 8 32" "2048 16 32" "400 6 22" "1024 8 8"; do
    echo "div=$div: $(GPRY_SK_DIV=$div timeout 300 python3 tools/r04/prof_lml_batch.py $shape 10 2>&1 | tail -1)" | tee -a gpurun_out/r05/sk_div.log
  done
done
