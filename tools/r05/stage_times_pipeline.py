#!/usr/bin/env python3
"""Stage timers of one LML + gradient evaluation with / without V = L^-1 underneath potrf: python3 tools/r05/stage_times_pipeline.py N ...
(GPRY_PANEL_FLAGS=16 / 32: the compact / the roomy form of the Cholesky step for every launch)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
for N in [int(a) for a in sys.argv[1:]]:
    d = 16 if N <= 4096 else 20
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    for pipe in (0, 1):
        dev.set_option("factor_pipeline", pipe)
        dev.set_option("lml_cache", 0)
        dev.lml(theta, True)
        dev.timing_reset()
        for _ in range(5):
            dev.lml(theta + 1e-9 * rng.standard_normal(d + 1), True)
        t = {k: dev.timing(k) for k in ("potrf", "trtri", "lauum", "lml_traces", "kernel_build")}
        print(f"N={N} pipeline={pipe}: " + ", ".join(f"{k} {v[0] / max(v[1], 1) * 1e3:.0f} us" for k, v in t.items()), flush=True)
        dev.set_option("timing", 0)
