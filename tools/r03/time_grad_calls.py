#!/usr/bin/env python3
"""One point: predict(std) + predict_grad (two device calls, what GaussianProcessRegressor.predict with gradients did) vs
predict_grad_batch of one row (one call)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
for N, d in ((64, 2), (256, 4), (1024, 8), (4096, 16)):
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1)
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-6)); dev.set_theta(3, np.log(np.array([2.0] + [0.4] * d)))
    assert dev.factorize() == 0
    x = rng.uniform(size=(1, d))
    def t(fn, reps=300):
        for _ in range(20): fn()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps * 1e6
    a = t(lambda: (dev.predict(x, return_std=True), dev.predict_grad(x[0])))
    b = t(lambda: dev.predict_grad_batch(x))
    c = t(lambda: dev.predict_grad_batch(x, want_kinv=False))
    p1 = t(lambda: dev.predict_point(x[0]))
    p0 = t(lambda: dev.predict_point(x[0], want_kinv=False))
    m1, s1 = dev.predict(x, return_std=True); mg1, kg1 = dev.predict_grad(x[0])
    m2, s2, mg2, kg2 = dev.predict_grad_batch(x)
    print(f"N={N} d={d}: ONE-POINT CALL {p1:.1f} us (mean gradient only {p0:.1f}); two calls {a:.1f} us, batch of one {b:.1f} us (mean gradient only {c:.1f}); |dmean| {abs(m1[0] - m2[0]):.1e} |dstd| {abs(s1[0] - s2[0]):.1e} "
          f"|dmg| {np.max(np.abs(mg1 - mg2[0])):.1e} |dkg| {np.max(np.abs(kg1 - kg2[0])):.1e}", flush=True)
    dev.close()
