#!/usr/bin/env python3
"""The cross-kernel panel of chunk c + 1 underneath the contraction of chunk c ("sweep_overlap"): same bits? faster?
ab_sweep_overlap.py [N d M]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N); Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
res = {}
for fresh in (0, 1):
    for ov in (0, 1, 0, 1):
        dev.set_option("sweep_overlap", ov)
        out = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("y", "sigma", "acq"))
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            dev.sweep_logexp(Xc if fresh else None, 0.1, 0.0, 1e-2, M=M, want=())
            ts.append(time.perf_counter() - t0)
        res.setdefault(ov, out)
        same = all(np.array_equal(out[k], res[0][k]) for k in ("y", "sigma", "acq"))
        print(f"N={N} M={M} {'fresh pool' if fresh else 'resident pool'} sweep_overlap={ov}: {min(ts) * 1e3:.2f} ms per sweep; bits equal to overlap=0: {same}", flush=True)
dev.close()
