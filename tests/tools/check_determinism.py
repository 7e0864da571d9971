#!/usr/bin/env python3
"""Run-to-run determinism and accuracy of the device factorisation at a given size."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib  # noqa: E402
from oracle import gpry_oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d))
y = rng.standard_normal(N)
alpha = np.full(N, 1e-4)
theta = np.log(np.array([4.0] + [0.3] * d))
dev = _lib.Device(0)
dev.set_train(X, y, alpha)
dev.set_theta(3, theta)
ref = orc.log_marginal_likelihood(X, y, alpha, theta, 3)
print("oracle lml", repr(float(ref)))
for la in (0, 1, 0, 1):
    dev.set_option("chol_lookahead", la)
    Ls, lmls = [], []
    for rep in range(3):
        assert dev.factorize() == 0
        Ls.append(np.tril(dev.get_factor(want_V=False, want_alpha=False)[0]))
        lmls.append(dev.lml(theta, False)[0])
    K = dev.kernel_train(add_alpha=True)
    err = np.max(np.abs(Ls[0] @ Ls[0].T - K)) / np.max(np.abs(K))
    print(f"lookahead={la}: repeat-identical {[bool(np.array_equal(Ls[0], L)) for L in Ls[1:]]} "
          f"|LL^T-K|/|K| = {err:.2e}; lml {[repr(v) for v in lmls]}")
    if not np.array_equal(Ls[0], Ls[1]):
        bad = np.argwhere(Ls[0] != Ls[1])
        print("   first differing entries (row, col):", bad[:5].tolist(), " count", len(bad),
              " min row", bad[:, 0].min(), " min col", bad[:, 1].min())
