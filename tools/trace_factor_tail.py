#!/usr/bin/env python3
"""A few LML+gradient evaluations at N=4096 for `rocprofv3 --kernel-trace`: what runs between the last panel step
of potrf and the trace kernel (the exposed tail of the factor chain)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 16
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d))
dev = _lib.Device(0)
dev.set_train(X, rng.standard_normal(N), np.full(N, 1e-4))
theta = np.log(np.array([4.0] + [0.3] * d))
dev.set_theta(3, theta)
for _ in range(6):
    dev.lml(theta, True)
