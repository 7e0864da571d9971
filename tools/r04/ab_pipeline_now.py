#!/usr/bin/env python3
"""Is the pipelined factor chain (V = L^-1 underneath potrf on a second stream, default from Np = 4096 until round 4, 1280 since) still ahead of the
serial chain with round 4's shorter panel steps?  Wall clock of one LML + gradient evaluation and of one factorisation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
dev.set_option("factor_pipeline_min", 0)
for N in [int(a) for a in (sys.argv[1:] or ["2048", "3072", "4096", "5120", "6144", "7168", "8192"])]:
    d = 8 if N < 4096 else (16 if N < 8192 else 20)
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    row = []
    for pipe in (0, 1):
        dev.set_option("factor_pipeline", pipe)
        for _ in range(3):
            dev.factorize(); dev.lml(theta, True)
        best_l = best_f = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                dev.lml(theta + 1e-9 * rep, True)
            best_l = min(best_l, (time.perf_counter() - t0) / 10 * 1e3)
            t0 = time.perf_counter()
            for _ in range(10):
                dev.set_theta(3, theta + 1e-9 * (rep + 1)); dev.factorize()
            best_f = min(best_f, (time.perf_counter() - t0) / 10 * 1e3)
        row.append(f"pipeline={pipe}: lml+grad {best_l:.3f} ms, factorize {best_f:.3f} ms")
    print(f"N={N}: " + " | ".join(row), flush=True)
