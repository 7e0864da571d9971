#!/bin/bash
# potrf at N=4096: where a panel step spends its time (section stamps) and the launch trace (durations, gaps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_q; mkdir -p $O
timeout 300 python3 tools/panel_sections.py 1 > $O/panel_sections.log 2>&1
cat > /tmp/potrf_only.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from gpry_amd import _lib
N, d = 4096, 16
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d))
dev = _lib.Device(0)
dev.set_option("factor_pipeline", 0)
dev.set_train(X, rng.standard_normal(N), np.full(N, 1e-4))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
for _ in range(5): assert dev.factorize() == 0
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 /tmp/potrf_only.py > $O/trace.log 2>&1
python3 tools/trace_fused.py $O/trace 64 > $O/trace_fused.log 2>&1
rm -rf $O/trace
cat $O/panel_sections.log $O/trace_fused.log
