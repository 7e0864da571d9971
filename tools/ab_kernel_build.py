"""Covariance build: round-2 kernel (kb_variant 0) vs the register-mirrored variants (1: 64 x 64 tiles, 2: 32 x 32
single-wave tiles).  Per variant: us per launch back to back (50 launches, gpry_microbench kind 6), us per launch
between two HIP events around ONE launch inside an LML evaluation (what bench.py's kernel_build object reports), and
the largest deviation of K from variant 0 relative to C (the F1 tolerance is 1e-13)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpry_amd import _lib

dev = _lib.Device(0)
print("| N | d | kernel | variant | us back to back | TB/s | frac of 8 TB/s | us by events (single launch) | max dK / C vs variant 0 |")
print("|---|---|---|---|---|---|---|---|---|")
for N, d, kid in ((4096, 16, 3), (8192, 20, 3), (1024, 8, 0), (2048, 5, 1)):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d)); y = rng.standard_normal(N)
    theta = np.log(np.array([4.0] + [0.3] * d))
    dev.set_train(X, y, 1e-4); dev.set_theta(kid, theta)
    K0 = None
    for var in (0, 1, 3):
        dev.set_option("kb_variant", var)
        us = min(dev.microbench(6, 50) for _ in range(3))
        dev.timing_reset()
        for _ in range(6): dev.lml(theta, False)
        ms, cnt = dev.timing("kernel_build")
        K = dev.kernel_train(add_alpha=True) if N <= 4096 else None
        dk = ""
        if K is not None:
            if K0 is None: K0 = K
            assert np.array_equal(K, K.T)
            dk = "%.1e" % (np.max(np.abs(K - K0)) / np.exp(theta[0]))
        nbytes = 8.0 * N * N + 8.0 * N * d
        print(f"| {N} | {d} | {kid} | {var} | {us:.1f} | {nbytes / us / 1e6:.2f} | {nbytes / us / 1e6 / 8:.3f} | {ms / max(cnt, 1) * 1e3:.1f} | {dk} |", flush=True)
    dev.set_option("kb_variant", 0)
