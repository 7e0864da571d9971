#!/usr/bin/env python3
"""K^-1 = V^T V with forced split-K factors (lauum_split) at the BASELINE sizes."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

dev = _lib.Device(0)
for N, d in ((2048, 12), (4096, 16), (8192, 20)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    ref = None
    for ns, lds in ((0, 0), (1, 32768), (2, 32768), (4, 32768), (2, 0), (0, 0), (2, 32768)):
        dev.set_option("lauum_split", ns)
        dev.set_option("lauum_lds", lds)
        out = dev.lml(theta, True); dev.timing_reset()
        for _ in range(5): out = dev.lml(theta, True)
        if ref is None: ref = out[1]
        print(f"N={N} lauum_split={ns} lds={lds}: lauum {dev.timing('lauum')[0] / 5:.3f} ms, trtri {dev.timing('trtri')[0] / 5:.3f} ms; "
              f"grad dev {np.max(np.abs(out[1] - ref)) / np.max(np.abs(ref)):.1e}", flush=True)
    dev.set_option("lauum_split", 0)
    dev.set_option("lauum_lds", 0)
