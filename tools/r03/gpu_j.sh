#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03_j; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; grep -E "passed|failed" $O/pytest_all.log
timeout 1200 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 600 python3 tools/latency_small_n.py > $O/latency_small_n.md 2>&1; cat $O/latency_small_n.md | head -20
