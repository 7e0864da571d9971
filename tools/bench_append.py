#!/usr/bin/env python3
"""Fixed-theta append (the "lie" of the acquisition step): bordered update (gpry_append_rows) against the
reference's route (send the enlarged training set, rebuild K, factorise, V = L^-1), same device, same sizes.
Also m points with x-gradients in one call against m single-point calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402


def med(fn, reps=7):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2] * 1e3


dev = _lib.Device(0)
for N, d in ((1000, 8), (4080, 16), (8172, 20)):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N + 64, d)); y = np.sin(3 * X).sum(1); a = np.full(N + 64, 1e-4)
    th = np.log(np.array([4.0] + [0.3] * d))
    for k in (1, 16, 64):
        def border():
            dev.set_train(X[:N], y[:N], a[:N]); dev.set_theta(3, th); assert dev.factorize() == 0
            t0 = time.perf_counter()
            assert dev.append_rows(X[N:N + k], y[N:N + k], a[N:N + k]) == 0
            border.t = time.perf_counter() - t0
        def full():
            t0 = time.perf_counter()
            dev.set_train(X[:N + k], y[:N + k], a[:N + k]); dev.set_theta(3, th); assert dev.factorize() == 0
            full.t = time.perf_counter() - t0
        tb, tf = [], []
        for _ in range(5):
            border(); tb.append(border.t); full(); tf.append(full.t)
        print(f"N={N} d={d} +{k:2d} rows: bordered update {sorted(tb)[2] * 1e3:7.3f} ms   rebuild + refactorise {sorted(tf)[2] * 1e3:7.3f} ms", flush=True)
    dev.set_train(X[:N], y[:N], a[:N]); dev.set_theta(3, th); assert dev.factorize() == 0
    for m in (16, 128):
        Xq = rng.uniform(0, 1, (m, d))
        tb = med(lambda: dev.predict_grad_batch(Xq, True))
        t1 = med(lambda: [dev.predict_grad(x, want_kinv=True) for x in Xq[:16]]) * m / 16
        print(f"N={N} d={d} gradients of {m:3d} points: one call {tb:7.3f} ms   single-point calls {t1:7.3f} ms", flush=True)
