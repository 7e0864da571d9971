#!/usr/bin/env python3
"""Small driver for rocprofv3: a few factorisations / LML evaluations at BASELINE sizes."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d))
y = rng.standard_normal(N)
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4))
theta = np.log(np.array([4.0] + [0.3] * d))
dev.set_theta(3, theta)
dev.factorize()
L0 = dev.get_factor()[0]
for ov in ((1, 0) if len(sys.argv) > 4 else (1,)):      # 4th argument: also the schedule with separate trailing launches
    dev.set_option("chol_overlap", ov)
    dev.timing_reset()
    for _ in range(reps):
        assert dev.factorize() == 0
        lml = dev.lml(theta, True)
    same = np.array_equal(np.tril(dev.get_factor()[0]), np.tril(L0))
    print(f"chol_overlap={ov}: factor bit-identical to the first one: {same}; lml {lml[0]:.12g}")
    for k in ("kernel_build", "potrf", "trtri", "lauum", "lml_traces"):
        ms, n = dev.timing(k)
        print(f"  {k}: {ms / max(n, 1) * 1e3:.1f} us avg over {n}")
dev.set_option("chol_overlap", 1)
