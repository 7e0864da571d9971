#!/bin/bash
# the N > 1 bench path on the one GPU of the box (two ranks sharing it: not a scaling figure) + a longer fuzz
mkdir -p gpurun_out/r06
{
GPRY_HIP_DEVICE_WRAP=1 timeout 900 python3 bench.py --gpus 2 --allow-gloo --steps 2 --warmup 1 > gpurun_out/r06/bench_2rank_one_gpu_gloo.json 2> gpurun_out/r06/bench_2rank_one_gpu_gloo.err
echo "2 ranks, one GPU, --allow-gloo: rc $?"; tail -c 600 gpurun_out/r06/bench_2rank_one_gpu_gloo.json | head -c 600; echo
GPRY_HIP_DEVICE_WRAP=1 timeout 600 python3 bench.py --gpus 2 --steps 2 --warmup 1 > /dev/null 2> gpurun_out/r06/bench_2rank_one_gpu_strict.err
echo "2 ranks, one GPU, strict: rc $? (3 = no RCCL communicator over 2 ranks on one device, as it must be)"; tail -2 gpurun_out/r06/bench_2rank_one_gpu_strict.err | cut -c1-300
for seed in 71 72 73; do timeout 1500 python3 tests/tools/fuzz_parity.py 300 $seed 2>&1 | tail -2; done
} 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r06/tworank_fuzz.log
