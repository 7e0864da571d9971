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
for blk, label in ((1, "diagonal workgroup"), (2, "first off-diagonal workgroup"), (9, "workgroup 8")):
    dev.set_option("chol_dbg", blk)
    dev.read_diag(True)
    assert dev.factorize() == 0
    dg = dev.read_diag(True).astype(float)
    dev.set_option("chol_dbg", 0)
    print(f"{label}: " + ", ".join(f"{n} {dg[i] / dg[5] / 100.0:.2f} us" for i, n in enumerate(names)) +
          f"  (s_memtime ticks of 10 ns; {int(dg[5])} panel steps)")

dev.set_option("chol_dbg", -1)      # dbg_block = -2: inside the 64x64 factor of the diagonal workgroup
dev.read_diag(True)
assert dev.factorize() == 0
dg = dev.read_diag(True).astype(float)
dev.set_option("chol_dbg", 0)
print(f"inside the 64x64 factor (cycles per panel step): chol16 x4 {dg[0] / dg[5]:.0f}, row solves x3 {dg[1] / dg[5]:.0f}, "
      f"MFMA updates x3 {dg[2] / dg[5]:.0f}")
