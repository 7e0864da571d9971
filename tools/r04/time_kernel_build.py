#!/usr/bin/env python3
"""Covariance build back to back (gpry_microbench kind 6): us per launch and fraction of 8 TB/s at N = 1024 ... 8192."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
for N, d, kid in ((1024, 8, 0), (2048, 16, 3), (4096, 16, 3), (4096, 16, 0), (8192, 20, 3)):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d))
    dev.set_train(X, rng.standard_normal(N), np.full(N, 1e-4)); dev.set_theta(kid, np.log(np.array([4.0] + [0.3] * d)))
    us = min(dev.microbench(6, 50) for _ in range(3))
    b = 8.0 * N * d + 8.0 * N * N
    print(f"N={N} d={d} kid={kid}: {us:.1f} us per launch = {b / us / 1e6:.2f} TB/s = {b / us / 1e6 / 8:.3f} of 8 TB/s", flush=True)
