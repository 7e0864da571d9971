#!/usr/bin/env python3
"""predict(return_std=True) for 1 / 8 / 64 points at N = 256 / 1024 under rocprofv3 --kernel-trace: which kernels make up the call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N, d = int(sys.argv[1]), 4
M = int(sys.argv[2])
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1)
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-6)); dev.set_theta(3, np.log(np.array([2.0] + [0.4] * d)))
assert dev.factorize() == 0
Xc = rng.uniform(size=(M, d))
for _ in range(20): dev.predict(Xc, return_std=True)
t0 = time.perf_counter()
for _ in range(200): dev.predict(Xc, return_std=True)
print(f"N={N} M={M}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call")
