#!/usr/bin/env python3
"""One shape of the batched objective for rocprofv3 --kernel-trace --stats: python3 tools/r04/prof_lml_batch.py N d B [reps] [kid]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
kid = int(sys.argv[5]) if len(sys.argv) > 5 else 3
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1); y = (y - y.mean()) / y.std()
dv = _lib.Device(0)
dv.set_train(X, y, np.full(N, 1e-4))
base = np.log(np.array([2.0] + [0.5] * d)); dv.set_theta(kid, base)
th = base + rng.uniform(-0.3, 0.3, (B, d + 1))
dv.lml_batch(th, True)
t0 = time.perf_counter()
for _ in range(reps):
    dv.lml_batch(th, True)
dt = (time.perf_counter() - t0) / reps
print(f"N={N} d={d} B={B}: {dt * 1e3:.3f} ms per call")
dv.close()
