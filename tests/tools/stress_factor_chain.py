#!/usr/bin/env python3
"""Stress: the pipelined factor chain (V = L^-1 phases on a second stream underneath potrf, stream-K launches) must
give the same bits every time -- a missing dependency between the two streams shows up as a different V, alpha or
LML gradient in some repetitions."""
import hashlib
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = _lib.Device(0)
bad = 0
for N, d in ((4096, 16), (5000, 8), (3000, 4), (6100, 6)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-5))
    theta = np.log(np.array([3.0] + [0.5] * d))
    dev.set_theta(3, theta)
    dev.set_option("factor_pipeline_min", 0)
    ref = None
    for r in range(reps):
        assert dev.factorize() == 0
        lml, grad, info = dev.lml(theta + 1e-3 * (r % 3), True)
        if r % 3:
            continue
        if r % 30 == 0:      # the whole factor now and then, the cheap fingerprints every time
            L, V, a = dev.get_factor()
            h = hashlib.sha1(V.tobytes() + a.tobytes()).hexdigest()
        key = (lml, grad.tobytes(), h)
        if ref is None:
            ref = key
        elif key != ref:
            bad += 1
            print(f"N={N} repetition {r}: differs from the first", flush=True)
    print(f"N={N}: {reps} repetitions done", flush=True)
dev.set_option("factor_pipeline_min", 1280)
print(f"{bad} deviations")
sys.exit(1 if bad else 0)
