#!/bin/bash
# round 3, first GPU call: new tests, self-launching bench, 1-GPU shard sizes for the predicted scaling table
O=gpurun_out/r03_a; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
GPRY_HIP_DEVICE_WRAP=1 timeout 900 python3 bench.py --gpus 2 --allow-gloo --steps 3 --warmup 1 > $O/bench_selflaunch_2rank.json 2> $O/bench_selflaunch_2rank.err; echo "selflaunch rc=$?"
timeout 300 python3 bench.py --gpus 2 --steps 3 > $O/bench_refuse.json 2> $O/bench_refuse.err; echo "refuse rc=$? (expect 2)"
for M in 1000000 500000 250000 125000; do
  timeout 900 python3 bench.py --steps 10 --warmup 3 --M $M --extras off --cpu-baseline off > $O/bench_M$M.json 2> $O/bench_M$M.err; echo "M=$M rc=$?"
done
GPRY_HIP_DEVICE_WRAP=1 timeout 1200 python3 bench.py --workload farm --mode group --gpus 2 --steps 1 --warmup 0 > $O/farm_group2.json 2> $O/farm_group2.err; echo "farm group rc=$?"
