#!/usr/bin/env python3
"""potrf stage time without checking the factor (timing experiments that break it: GPRY_PANEL_FLAGS=4): python3 tools/r05/time_potrf_raw.py N ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
dev.set_option("factor_pipeline", 0)
out = []
for N in [int(a) for a in sys.argv[1:]]:
    d = 8 if N < 4096 else 16
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1.0))          # a heavy diagonal: positive definite whatever is left out
    dev.set_theta(3, np.log(np.array([0.01] + [0.3] * d)))
    dev.factorize()
    best = None
    for rep in range(3):
        dev.timing_reset()
        for _ in range(5):
            dev.factorize()
        t = dev.timing("potrf"); ms = t[0] / max(t[1], 1)
        best = ms if best is None else min(best, ms)
    out.append(f"N={N}: potrf {best * 1e3:.0f} us")
print("; ".join(out))
