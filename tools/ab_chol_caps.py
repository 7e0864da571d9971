#!/usr/bin/env python3
"""Tile rounds per fused Cholesky launch (option chol_caps = 16 * first-step rounds + second-step rounds)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
dev = _lib.Device(0)
for N, d in ((2048, 8), (4096, 16), (6144, 20), (8192, 20)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    dev.set_option("chol_overlap_max", 8192)
    ref = None
    for pair, multi, caps in ((0, 2, 0x21), (1, 2, 0x21), (1, 2, 0x32), (1, 1, 0x21), (1, 3, 0x21)):
        dev.set_option("chol_pair", pair)
        dev.set_option("chol_multi", multi)
        dev.set_option("chol_caps", caps)
        assert dev.factorize() == 0
        L = np.tril(dev.get_factor(want_V=False, want_alpha=False)[0])
        if ref is None:
            ref = L
        dev.timing_reset()
        for _ in range(5):
            assert dev.factorize() == 0
        print(f"N={N} pair {pair} multi {multi} caps {caps >> 4},{caps & 15}: potrf {dev.timing('potrf')[0] / 5 * 1e3:7.1f} us  "
              f"bit-identical to the unpaired plan: {np.array_equal(L, ref)}", flush=True)
        dev.set_option("timing", 0)
