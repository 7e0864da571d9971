#!/usr/bin/env python3
"""Batched objective at the larger sizes: python3 tools/r04/time_batch_sched.py N d B ...   (ms per call, LML + gradient)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N, d = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1); y = (y - y.mean()) / y.std()
dv = _lib.Device(0)
dv.set_option("lml_batch", 8192)
dv.set_train(X, y, np.full(N, 1e-4))
base = np.log(np.array([2.0] + [0.5] * d)); dv.set_theta(3, base)
out = []
for B in [int(b) for b in sys.argv[3:]]:
    th = base + rng.uniform(-0.3, 0.3, (B, d + 1))
    dv.lml_batch(th, True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); dv.lml_batch(th, True); ts.append(time.perf_counter() - t0)
    out.append(f"B={B} {min(ts) * 1e3:.2f}")
print(f"N={N} d={d}: " + "; ".join(out))
dv.close()
