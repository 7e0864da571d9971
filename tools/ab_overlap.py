#!/usr/bin/env python3
"""A/B of the whole resident sweep (cross_build + contraction + finish) with and without the
two-stream overlap, interleaved in one process."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d, M = 4096, 16, 1000000
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N); Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
ref = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("acq",))["acq"]
res = {0: [], 1: []}
gem = {0: [], 1: []}
for rnd in range(5):
    for ov in (0, 1):
        dev.set_option("sweep_overlap", ov)
        out = dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=("acq",))
        assert np.array_equal(out["acq"], ref, equal_nan=True), ov
        dev.timing_reset()
        dev.sync()
        t0 = time.perf_counter()
        dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
        dev.sync()
        res[ov].append((time.perf_counter() - t0) * 1e3)
        ms, n = dev.timing("sweep_gemm")
        gem[ov].append(ms / n)
for ov in (0, 1):
    print(f"overlap={ov}: sweep wall ms median {np.median(res[ov]):.1f} (all {' '.join(f'{v:.1f}' for v in res[ov])}); "
          f"GEMM launch ms median {np.median(gem[ov]):.3f}")
