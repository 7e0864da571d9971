#!/usr/bin/env python3
"""Stress run for the kernels that synchronise through counters / LDS flags (Cholesky panel steps): many
factorisations at random sizes, alone and from three threads at once, every result checked."""
import os
import sys
import threading
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = []
count = [0] * n_threads


def work(k):
    dev = _lib.Device(0)
    dev.set_option("timing", 0)
    rng = np.random.default_rng(100 + k)
    t_end = time.time() + seconds
    while time.time() < t_end:
        N = int(rng.choice([60, 128, 129, 200, 500, 640, 1000, 1300, 2048, 2500]))
        d = int(rng.integers(1, 9))
        X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
        theta = np.log(np.append(rng.uniform(0.5, 5.0), rng.uniform(0.1, 1.0, d)))
        dev.set_train(X, y, np.full(N, 1e-5)); dev.set_theta(int(rng.integers(0, 4)), theta)
        for _ in range(3):
            if dev.factorize() != 0:
                bad.append(("info", N, d)); break
            lml, g, info = dev.lml(theta, True)
        L, V, a = dev.get_factor()
        K = dev.kernel_train(add_alpha=True)
        e1 = np.max(np.abs(L @ L.T - K)) / np.max(np.abs(K))
        e2 = np.max(np.abs(V @ L - np.eye(N)))
        if not (e1 < 1e-12 and e2 < 1e-6 and np.isfinite(lml)):
            bad.append((N, d, e1, e2, lml))
        count[k] += 1
    dev.close()


ths = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
for t in ths: t.start()
for t in ths: t.join()
print(f"{sum(count)} models x 3 factorisations + LML evaluations from {n_threads} thread(s) in {seconds:.0f} s; failures: {len(bad)}", bad[:3])
