#!/usr/bin/env python3
"""lml_traces stage of an LML + gradient evaluation (stage timers of the library): python3 tools/r05/time_traces.py N d [kid] ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
out = []
args = sys.argv[1:] or ["1024", "8", "4096", "16", "8192", "20"]
for N, d in zip(map(int, args[0::2]), map(int, args[1::2])):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    dev.lml(theta, True)
    best = None
    for rep in range(3):
        dev.timing_reset()
        for _ in range(5):
            dev.lml(theta, True)
        t = dev.timing("lml_traces"); ms = t[0] / max(t[1], 1)
        best = ms if best is None else min(best, ms)
    dev.set_option("timing", 0)
    out.append(f"N={N} d={d}: lml_traces {best * 1e3:.1f} us")
print("; ".join(out))
