#!/bin/bash
O=gpurun_out/r03_h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -4 $O/pytest_all.log | head -2
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python tools/latency_lml.py > $O/latency_lml.log 2>&1; cat $O/latency_lml.log
