#!/usr/bin/env python3
"""Cholesky with the trailing-update tiles riding in the panel launches (chol_overlap=1, default) against
the schedule with separate trailing launches (chol_overlap=0: every trailing update its own launch, segment by segment above Np = 3584): potrf time
and bit-identity of the factor, at several sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

dev = _lib.Device(0)
for N, d in [(int(a), 8 if int(a) < 4096 else (16 if int(a) < 8192 else 20)) for a in (sys.argv[1:] or ["256", "1024", "2048", "4096", "8192"])]:
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d))
    dev.set_theta(3, theta)
    out = {}
    for name, ov in (("separate launches", 0), ("tiles in panel launches", 1)):
        dev.set_option("chol_overlap", ov)
        assert dev.factorize() == 0
        dev.timing_reset()
        for _ in range(5):
            assert dev.factorize() == 0
            lml = dev.lml(theta, True)
        L = np.tril(dev.get_factor(want_V=False, want_alpha=False)[0])
        t = {k: dev.timing(k)[0] / max(dev.timing(k)[1], 1) * 1e3 for k in ("potrf", "trtri", "lauum")}
        out[name] = (L, t, lml[0])
        dev.set_option("timing", 0)
    ref = out["separate launches"][0]
    for name, (L, t, lml) in out.items():
        print(f"N={N:5d} {name:32s}: potrf {t['potrf']:8.1f} us  trtri {t['trtri']:7.1f}  lauum {t['lauum']:7.1f}  "
              f"bit-identical to separate launches: {np.array_equal(L, ref)}  lml {lml:.12g}", flush=True)
dev.set_option("chol_overlap", 1)
