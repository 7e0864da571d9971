#!/usr/bin/env python3
"""Device-side stage times of gpry_predict(return_std=True) for small batches (HIP events)."""
import os, sys, time
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
    for M in (17, 64, 400, 2000):
        Xc = rng.uniform(0, 1, (M, d))
        for _ in range(3):
            dev.predict(Xc, return_std=True)
        dev.timing_reset()
        t0 = time.perf_counter()
        for _ in range(20):
            dev.predict(Xc, return_std=True)
        wall = (time.perf_counter() - t0) / 20 * 1e6
        st = {k: dev.timing(k)[0] / 20 * 1e3 for k in ("cross_build", "sweep_gemm_splitk", "sweep_gemm", "sweep_finish")}
        dev.set_option("timing", 0)
        t0 = time.perf_counter()
        for _ in range(20):
            dev.predict(Xc, return_std=True)
        wall0 = (time.perf_counter() - t0) / 20 * 1e6
        print(f"N={N} M={M}: wall {wall0:.0f} us (with events {wall:.0f}); stages us: " +
              ", ".join(f"{k} {v:.1f}" for k, v in st.items()))
    dev.close()
