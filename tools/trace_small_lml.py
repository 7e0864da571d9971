#!/usr/bin/env python3
"""One LML+gradient evaluation at small N under `rocprofv3 --kernel-trace`: which kernels run and how
much of the wall clock lies between them (launch-bound regime)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 4
dev = _lib.Device(0)
rng = np.random.default_rng(N)
X = rng.uniform(0, 1, (N, d)); y = np.sin(3 * X).sum(1)
dev.set_train(X, y, np.full(N, 1e-6))
theta = np.log(np.array([2.0] + [0.4] * d)); dev.set_theta(3, theta)
for _ in range(20):
    dev.lml(theta, True)
t0 = time.perf_counter()
for i in range(100):
    dev.lml(theta + 1e-4 * (i % 7), True)
print(f"N={N}: lml+grad {(time.perf_counter() - t0) / 100 * 1e6:.1f} us per call")
