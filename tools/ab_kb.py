#!/usr/bin/env python3
"""A/B of the kernel-build tile size in one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
for (N, d) in ((4096, 16), (8192, 20), (1024, 8)):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-4))
    dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    ref = None
    for rnd in range(3):
        for ts in (64, 32):
            dev.set_option("kb_tile", ts)
            K = dev.kernel_train(True)
            if ref is None: ref = K
            assert np.array_equal(K, ref)
            dev.timing_reset()
            for _ in range(10): dev._lib.gpry_kernel_train(dev._h, 1, None)
            ms, n = dev.timing("kernel_build")
            print(f"N={N} d={d} TS={ts}: {ms/n*1e3:.1f} us -> {(8.0*N*N+8.0*N*d)/(ms/n*1e-3)/1e9:.0f} GB/s")
    dev.close()
