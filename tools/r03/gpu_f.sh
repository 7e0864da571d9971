#!/bin/bash
O=gpurun_out/r03_f; mkdir -p $O
timeout 300 python tools/lml_small_sections.py > $O/lml_small_sections.log 2>&1; cat $O/lml_small_sections.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc=$?"; tail -5 $O/pytest_all.log | head -3
timeout 900 python tools/latency_small_n.py > $O/latency_small_n.md 2>&1; cat $O/latency_small_n.md
