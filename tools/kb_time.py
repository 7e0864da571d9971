#!/usr/bin/env python3
"""Covariance build (kernel_train_kernel) at N = 4096 / 8192: microseconds per launch, 50 launches back to back."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
dev = _lib.Device(0)
for N, d in ((1024, 8), (4096, 16), (8192, 20)):
    rng = np.random.default_rng(0)
    dev.set_train(rng.uniform(0, 1, (N, d)), rng.standard_normal(N), np.full(N, 1e-4))
    dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    us = min(dev.microbench(6, 50) for _ in range(5))
    print(f"N={N}: {us:.2f} us per launch, {(8.0 * N * N + 8.0 * N * d) / (us * 1e-6) / 1e9:.0f} GB/s = {(8.0 * N * N + 8.0 * N * d) / (us * 1e-6) / 8e12:.3f} of 8 TB/s")
