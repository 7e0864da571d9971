#!/bin/bash
# potrf of the serial chain by size of the final (riding-tile) segment and width of the outer blocks (environment knobs of a tuning build)
mkdir -p gpurun_out/r05
for tail in 2048 2560 3072 3584 4096; do
  for blk in 512 768 1024; do
    echo -n "tail $tail block $blk: "
    GPRY_CHOL_TAIL=$tail GPRY_CHOL_BLOCK=$blk timeout 200 python3 tools/r04/time_potrf.py 3072 4096 5120 6144 8192 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/r05/sweep_segments.log
