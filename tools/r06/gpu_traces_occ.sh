#!/bin/bash
# lml_traces_kernel with five waves per SIMD instead of four (2080 workgroups of four waves at N = 4096: 2.03 rounds of 1024 slots):
# A the tree, B launch bound only (96 VGPRs + 64 B of scratch), D one coordinate in flight in the second pass (96 VGPRs, no scratch)
mkdir -p gpurun_out/r06
{
for rep in 1 2; do
for v in A B D; do
  lib=$PWD/gpry_amd/libgpry_hip.so; [ $v != A ] && lib=$PWD/gpry_amd/libgpry_hip_$v.so
  echo "variant $v: $(GPRY_HIP_LIB=$lib timeout 600 python3 tools/r05/time_traces.py 1024 8 2048 16 4096 16 4096 4 4096 8 4096 32 8192 20 2>&1 | tail -1)"
done
done
} 2>&1 | tee gpurun_out/r06/traces_occ.log
