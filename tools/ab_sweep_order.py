#!/usr/bin/env python3
"""Sweep contraction: super-tile order (row super-tile outermost vs candidate super-column outermost)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d, M = 4096, 16, 262144
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N); Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
ref = None
for order, tm in ((0, 3), (1, 3), (0, 3), (1, 3), (1, 2), (1, 4)):
    dev.set_option("sweep_colouter", order); dev.set_option("sweep_tilemap", tm)
    out = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("sigma",))
    dev.timing_reset()
    for _ in range(3): dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
    ms, n = dev.timing("sweep_gemm")
    if ref is None: ref = out["sigma"]
    print(f"colouter={order} tilemap={tm}: sweep_gemm {ms / n:.3f} ms per launch, {M * 3 * (N * N + 2.0 * N) / (ms * 1e-3) / 1e12:.2f} TF; "
          f"sigma identical: {np.array_equal(out['sigma'], ref)}", flush=True)
