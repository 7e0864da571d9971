#!/usr/bin/env python3
"""Beyond the benchmark sizes: one LML + gradient evaluation and a sweep at N = 12288 / 16384 against scipy's Cholesky of the
same matrix on the host (value) and a central difference (gradient), with timings.  Not a test (minutes of host LAPACK)."""
import os
import sys
import time
import numpy as np
import scipy.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib  # noqa: E402

d = 8
for N in (int(a) for a in (sys.argv[1:] or ["12288", "16384"])):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = np.sin(3 * X).sum(1) + 0.1 * rng.standard_normal(N)
    theta = np.log(np.array([2.0] + [0.35] * d))
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-3)); dev.set_theta(3, theta)
    t0 = time.perf_counter(); lml, grad, info = dev.lml(theta, True); t1 = time.perf_counter()
    lml2, grad2, info2 = dev.lml(theta, True); t2 = time.perf_counter()
    K = dev.kernel_train(add_alpha=True)
    c = scipy.linalg.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
    ref = -0.5 * y @ scipy.linalg.cho_solve(c, y, check_finite=False) - np.log(np.diag(c[0])).sum() - 0.5 * N * np.log(2 * np.pi)
    del K, c
    k = 1; h = 1e-5
    tp, tm = theta.copy(), theta.copy(); tp[k] += h; tm[k] -= h
    fd = (dev.lml(tp, False)[0] - dev.lml(tm, False)[0]) / (2 * h)
    M = 65536
    Xc = rng.uniform(0, 1, (M, d))
    dev.set_theta(3, theta); assert dev.factorize() == 0
    t3 = time.perf_counter(); out = dev.sweep_logexp(Xc, 0.2, float(y.max()), 1e-2, want=("y", "sigma")); t4 = time.perf_counter()
    m1, s1 = dev.predict(Xc[:500], return_std=True)
    print(f"N={N}: LML+grad first call {1e3 * (t1 - t0):.1f} ms, second {1e3 * (t2 - t1):.1f} ms; LML {lml:.6f} vs host Cholesky {ref:.6f} "
          f"(rel {abs(lml - ref) / abs(ref):.1e}); dLML/dtheta_1 {grad[k]:.6f} vs central difference {fd:.6f}; sweep of {M} candidates "
          f"{1e3 * (t4 - t3):.1f} ms; sweep vs predict: mean {np.max(np.abs(out['y'][:500] - m1)):.1e}, sigma {np.max(np.abs(out['sigma'][:500] - s1)):.1e}", flush=True)
    dev.close()
