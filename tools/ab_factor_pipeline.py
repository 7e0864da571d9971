#!/usr/bin/env python3
"""V = L^-1 queued underneath the Cholesky panel chain (factor_pipeline=1) against the serial chain:
wall-clock of one factorisation and of one LML+gradient evaluation, stage timers, bit-identity."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

dev = _lib.Device(0)
dev.set_option("factor_pipeline_min", 0)
for N, d in [(int(a), 8 if int(a) < 4096 else (16 if int(a) < 8192 else 20)) for a in (sys.argv[1:] or ["1024", "2048", "4096", "6144", "8192"])]:
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d))
    dev.set_theta(3, theta)
    ref = None
    for rep in range(2):
        for pipe, spine in ((0, 0), (1, 0), (0, -1), (1, -1)):
            dev.set_option("factor_pipeline", pipe)
            dev.set_option("gemm_streamk", 1 << 20 if spine < 0 else 0)
            for _ in range(3):
                assert dev.factorize() == 0
                dev.lml(theta, True)
            t0 = time.perf_counter()
            for _ in range(10):
                assert dev.factorize() == 0
            tf = (time.perf_counter() - t0) / 10 * 1e3
            t0 = time.perf_counter()
            for _ in range(10):
                lml = dev.lml(theta, True)
            tl = (time.perf_counter() - t0) / 10 * 1e3
            dev.timing_reset()
            for _ in range(5):
                dev.lml(theta, True)
            t = {k: dev.timing(k)[0] / max(dev.timing(k)[1], 1) * 1e3 for k in ("potrf", "trtri", "lauum")}
            dev.set_option("timing", 0)
            V = dev.get_factor(want_alpha=False)[1]
            if ref is None:
                ref = V
            print(f"N={N:5d} factor_pipeline={pipe} stream-K {int(spine < 0)}: factorize {tf:7.3f} ms  lml+grad {tl:7.3f} ms   stage timers us: "
                  f"potrf {t['potrf']:8.1f} trtri {t['trtri']:7.1f} lauum {t['lauum']:7.1f}   V bit-identical to the first: {np.array_equal(V, ref)}  max |dV| {np.max(np.abs(V - ref)):.1e}", flush=True)
dev.set_option("factor_pipeline", 1)
dev.set_option("factor_pipeline_min", 1280)
dev.set_option("gemm_streamk", 5632)
