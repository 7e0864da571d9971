#!/usr/bin/env python3
"""The stacked inverse factor must not depend on what the scratch matrices held before: factorise, poison dW / dW2 / dW3 through an
objective evaluation at another size pattern, factorise again -- and the same with NaN-filled scratch via a larger model first."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
dev = _lib.Device(0)
rng = np.random.default_rng(0)
for N, d in ((129, 2), (200, 3), (500, 4), (1000, 5), (2048, 6)):
    # a larger, ill-conditioned model first: its factor fills the buffers with large numbers (and its padding differs)
    Nb = N + 300
    Xb = rng.uniform(size=(Nb, d)); yb = rng.standard_normal(Nb)
    dev.set_train(Xb, yb, np.full(Nb, 1e-9)); dev.set_theta(0, np.log(np.array([1e6] + [5.0] * d))); dev.factorize(); dev.lml(np.log(np.array([1e6] + [5.0] * d)), True)
    X = Xb[:N]; y = yb[:N]
    dev.set_train(X, y, np.full(N, 1e-5)); th = np.log(np.array([2.0] + [0.4] * d)); dev.set_theta(3, th)
    assert dev.factorize() == 0
    L, V, a = dev.get_factor()
    Vn = np.linalg.inv(np.tril(L))
    l = dev.lml(th, True)
    assert dev.factorize() == 0
    L2, V2, a2 = dev.get_factor()
    print(f"N={N}: V vs numpy {np.abs(np.tril(V) - Vn).max() / np.abs(Vn).max():.1e}; repeat identical {np.array_equal(V, V2) and np.array_equal(a, a2)}; finite {np.isfinite(V).all() and np.isfinite(l[1]).all()}")
