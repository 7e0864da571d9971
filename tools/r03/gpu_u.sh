#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_lml_small_gpu.py tests/test_host_mirror_gpu.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python3 tools/lml_small_sections.py 2>&1 | tail -8
timeout 300 python3 tools/latency_lml.py 2>&1 | tail -8
