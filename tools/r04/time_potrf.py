#!/usr/bin/env python3
"""potrf / V = L^-1 / K^-1 stage times of the serial factor chain (factor_pipeline = 0) at several sizes: python3 tools/r04/time_potrf.py N ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
dev.set_option("factor_pipeline", 0)
for kv in filter(None, os.environ.get("GPRY_SET", "").split(",")):       # e.g. GPRY_SET=chol_lookahead=0
    dev.set_option(kv.split("=")[0], int(kv.split("=")[1]))
out = []
for N in [int(a) for a in (sys.argv[1:] or ["512", "1024", "2048", "4096", "6144", "7168"])]:
    d = 8 if N < 4096 else 16
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    assert dev.factorize() == 0
    best = None
    for rep in range(3):
        dev.timing_reset()
        for _ in range(5):
            assert dev.factorize() == 0
        t = dev.timing("potrf"); ms = t[0] / max(t[1], 1)
        best = ms if best is None else min(best, ms)
    dev.set_option("timing", 0)
    out.append(f"N={N}: potrf {best * 1e3:.0f} us")
print("; ".join(out))
