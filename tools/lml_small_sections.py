"""Cycles per phase of the single-launch objective (lml_small.hip), thread 0's clock: B build, C Cholesky chain,
E diagonal inverses, F V = L^-1, G vectors, H traces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpry_amd import _lib
dev = _lib.Device(0)
for N, d in ((32, 2), (64, 2), (128, 4), (128, 16), (100, 8)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = np.sin(3 * X).sum(1)
    dev.set_train(X, y, np.full(N, 1e-6))
    theta = np.log(np.array([2.0] + [0.4] * d)); dev.set_theta(3, theta)
    for _ in range(5): dev.lml(theta, True)
    dev.set_option("chol_dbg", 1)
    dev.read_diag(reset=True)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps): dev.lml(theta, True)
    wall = (time.perf_counter() - t0) / reps * 1e6
    c = dev.read_diag(reset=True).astype(float) / reps
    dev.set_option("chol_dbg", 0)
    t0 = time.perf_counter()
    for _ in range(200): dev.lml(theta, True)
    wall0 = (time.perf_counter() - t0) / 200 * 1e6
    print(f"N={N:4d} d={d:2d}: build {c[0]:7.0f}  chol {c[1]:7.0f}  diag-inv {c[2]:7.0f}  V {c[3]:7.0f}  vectors {c[4]:7.0f}  traces {c[5]:7.0f}"
          f"  sum {c.sum():8.0f} cycles = {c.sum() / 2400:.1f} us at 2.4 GHz; wall {wall0:.1f} us per call", flush=True)
