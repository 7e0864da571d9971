#!/usr/bin/env python3
"""V = L^-1 with the split-K factor of its levels capped (trtri_split_cap)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

dev = _lib.Device(0)
for N, d in ((1024, 8), (2048, 12), (4096, 16), (8192, 20)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    for cap in (0, 1, 2, 4, 0, 2):
        dev.set_option("trtri_split_cap", cap)
        dev.factorize(); dev.timing_reset()
        for _ in range(6): dev.factorize()
        print(f"N={N} trtri_split_cap={cap}: trtri {dev.timing('trtri')[0] / 6:.3f} ms", flush=True)
    dev.set_option("trtri_split_cap", 0)
