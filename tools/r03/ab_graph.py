#!/usr/bin/env python3
"""Covariance build + potrf (serial chain) launched kernel by kernel vs replayed as a captured hipGraph (gpry_microbench kind 8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
for N, d in ((256, 4), (512, 4), (1024, 8), (2048, 12), (4096, 16)):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d)); y = rng.standard_normal(N)
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    res = []
    for rep in range(2):
        for mode in (0, 1):
            res.append((mode, dev.microbench(8, (40 << 1) | mode)))
    print(f"N={N}: " + ", ".join(f"{'graph' if m else 'launches'} {v:.1f} us" for m, v in res), flush=True)
    dev.close()
