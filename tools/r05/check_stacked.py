#!/usr/bin/env python3
"""The inverse factor as extra rows of the panel chain (option chol_stacked) against numpy and against the recursive inverse of the
same library, and the time of factorize / LML + gradient with and without it: python3 tools/r05/check_stacked.py N ..."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
def best(f, reps=7):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
for N in [int(a) for a in (sys.argv[1:] or ["200", "256", "1000", "1024", "2048", "3072"])]:
    d = 8
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4)); theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    out = {}
    for mode in (0, 4096):
        dev.set_option("chol_stacked", mode)
        assert dev.factorize() == 0
        L, V, a = dev.get_factor()
        lml = dev.lml(theta, True)
        out[mode] = (L, V, a, lml, best(lambda: dev.factorize()), best(lambda: dev.lml(theta, True)))
    L0, V0, a0, l0, tf0, tl0 = out[0]; L1, V1, a1, l1, tf1, tl1 = out[4096]
    Vn = np.linalg.inv(np.tril(L1)[:N, :N])
    eV = np.abs(np.tril(V1)[:N, :N] - Vn).max() / np.abs(Vn).max()
    eV0 = np.abs(np.tril(V0)[:N, :N] - Vn).max() / np.abs(Vn).max()
    up = np.abs(np.triu(V1, 1)).max()
    print(f"N={N}: L identical {np.array_equal(L0, L1)}; V vs numpy: stacked {eV:.1e}, recursive {eV0:.1e}; above the diagonal {up:.1e}; "
          f"alpha rel diff {np.abs(a1 - a0).max() / np.abs(a0).max():.1e}; lml diff {abs(l1[0] - l0[0]) / abs(l0[0]):.1e}, grad diff {np.abs(l1[1] - l0[1]).max() / np.abs(l0[1]).max():.1e} | "
          f"factorize {tf0:.3f} -> {tf1:.3f} ms, lml+grad {tl0:.3f} -> {tl1:.3f} ms")
dev.set_option("chol_stacked", 2048)
