#!/usr/bin/env python3
"""Wall-clock latency of one LML+gradient evaluation and one factorisation at small and medium N
(the regime in which an active-learning run spends most of its iterations)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

dev = _lib.Device(0)
timing = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev.set_option("timing", timing)
for N, d in ((32, 2), (64, 2), (128, 4), (256, 4), (512, 8), (1024, 8), (2048, 12)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = np.sin(3 * X).sum(1)
    dev.set_train(X, y, np.full(N, 1e-6))
    theta = np.log(np.array([2.0] + [0.4] * d)); dev.set_theta(3, theta)
    for _ in range(3): dev.lml(theta, True); dev.factorize()
    reps = 200 if N <= 512 else 50
    t0 = time.perf_counter()
    for i in range(reps): dev.lml(theta + 1e-4 * (i % 7), True)
    t1 = time.perf_counter()
    for i in range(reps): dev.lml(theta + 1e-4 * (i % 7), False)
    t2 = time.perf_counter()
    for i in range(reps): dev.factorize()
    t3 = time.perf_counter()
    print(f"N={N:5d} d={d:2d}: lml+grad {(t1 - t0) / reps * 1e6:8.1f} us, lml only {(t2 - t1) / reps * 1e6:8.1f} us, "
          f"factorize {(t3 - t2) / reps * 1e6:8.1f} us", flush=True)
