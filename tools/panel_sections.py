#!/usr/bin/env python3
"""Where a workgroup of the fused Cholesky panel step spends its time (s_memtime section stamps)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d = 4096, 16
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d))
dev = _lib.Device(0)
dev.set_train(X, rng.standard_normal(N), np.full(N, 1e-4))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
names = ["loads", "left-looking update", "64x64 factor", "row solve / wait", "store"]
overlap = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev.set_option("chol_overlap", overlap)
print(f"chol_overlap = {overlap} (s_memtime ticks are 100 MHz: x24 for core cycles at 2.4 GHz)")
for blk, label in ((1, "diagonal workgroup"), (2, "first off-diagonal workgroup"), (9, "workgroup 8")):
    dev.set_option("chol_dbg", blk)
    dev.read_diag(True)
    assert dev.factorize() == 0
    dg = dev.read_diag(True).astype(float)
    dev.set_option("chol_dbg", 0)
    print(f"{label}: " + ", ".join(f"{n} {dg[i] / dg[5]:.0f}" for i, n in enumerate(names)) +
          f"  (s_memtime cycles per panel step; {int(dg[5])} steps)")
