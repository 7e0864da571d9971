#!/usr/bin/env python3
"""Wall-clock latency of gpry_predict for small batches (nested-sampler / MCMC call pattern)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

for N, d in ((1024, 8), (4096, 16)):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-4))
    dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    assert dev.factorize() == 0
    if len(sys.argv) > 1:
        dev.set_option("timing", int(sys.argv[1]))
    for M in (1, 16, 17, 64, 400, 2000):
        Xc = rng.uniform(0, 1, (M, d))
        for std in (False, True):
            for _ in range(5):
                dev.predict(Xc, return_std=std)
            t0 = time.perf_counter()
            reps = 50
            for _ in range(reps):
                dev.predict(Xc, return_std=std)
            us = (time.perf_counter() - t0) / reps * 1e6
            line = f"N={N} d={d} M={M:5d} std={int(std)}: {us:8.1f} us per call, {us / M:8.2f} us per point"
            if std and M > 16:      # the same call through the one-pass contraction (round 1's path)
                dev.set_option("predict_split", 0)
                for _ in range(3):
                    dev.predict(Xc, return_std=True)
                t0 = time.perf_counter()
                for _ in range(20):
                    dev.predict(Xc, return_std=True)
                line += f"   (one-pass contraction: {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us)"
                dev.set_option("predict_split", 1)
            print(line)
    dev.close()

# one point with x-gradients (what a gradient-based acquisition optimiser calls per step)
for N, d in ((1024, 8), (4096, 16)):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-4))
    dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    assert dev.factorize() == 0
    x = rng.uniform(0, 1, d)
    for kinv in (False, True):
        for _ in range(5):
            dev.predict_grad(x, want_kinv=kinv)
        t0 = time.perf_counter()
        for _ in range(50):
            dev.predict_grad(x, want_kinv=kinv)
        print(f"N={N} d={d} predict_grad(want_kinv={kinv}): {(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
    dev.close()
