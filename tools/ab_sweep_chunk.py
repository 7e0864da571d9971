#!/usr/bin/env python3
"""Sweep of M = 1e6 candidates at N=4096, d=16 for several chunk sizes (candidates per launch of the
contraction): bigger chunks mean fewer launch tails (tiles of a launch differ 32x in length)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gpry_amd import _lib

N, d, M = 4096, 16, 1_000_000
bounds, X, y, Xc, _ = bench.synthetic(N, d, M)
dev = _lib.Device(0)
lo, span = bounds[:, 0], bounds[:, 1] - bounds[:, 0]
X_ = (X - lo) / span
ys = (y - y.mean()) / y.std()
dev.set_train(X_, ys, np.full(N, 1e-4 / y.std() ** 2))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
dev.set_affine(lo, span, y.mean(), y.std(), np.inf)
chunks = [int(v) for v in sys.argv[1:]] or [32768, 65536, 131072, 262144]
dev.sweep_logexp(Xc, 0.1, 0.0, 0.01, want=())
for ch in chunks + chunks[:1]:
    dev.set_option("sweep_chunk", ch)
    dev.sweep_logexp(None, 0.1, 0.0, 0.01, M=M, want=())
    dev.timing_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.sweep_logexp(None, 0.1, 0.0, 0.01, M=M, want=())
    wall = (time.perf_counter() - t0) / 3 * 1e3
    g, n = dev.timing("sweep_gemm")
    c, _ = dev.timing("cross_build")
    print(f"chunk {ch:7d}: sweep {wall:7.2f} ms wall, contraction {g / 3:7.2f} ms in {n // 3} launches, panel build {c / 3:6.2f} ms, "
          f"{M * (N * N + 2.0 * N) / (g / 3 * 1e-3) / 1e12:.2f} TFLOP/s", flush=True)
    dev.set_option("timing", 0)
