#!/usr/bin/env python3
"""Factor chain (L = chol(K), V = L^-1) hand-written vs rocSOLVER (dpotrf + dtrtri) on the same
matrices: the measurement behind the choice BASELINE.json's north_star asks to justify."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = _lib.Device(0)
for N, d in ((1024, 8), (2048, 12), (4096, 16), (8192, 20)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d))
    y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    out = {}
    for name, opt in (("hand-written", 0), ("rocSOLVER", 1)):
        dev.set_option("chol", opt)
        assert dev.factorize() == 0          # warm-up (rocSOLVER: dlopen, workspace)
        dev.timing_reset()
        for _ in range(reps):
            assert dev.factorize() == 0
        ms = sum(dev.timing(k)[0] for k in ("potrf", "trtri")) / reps
        L, V, _ = dev.get_factor()
        out[name] = (ms, L, V)
    dev.set_option("chol", 0)
    (t0, L0, V0), (t1, L1, V1) = out["hand-written"], out["rocSOLVER"]
    flops = 2.0 * N ** 3 / 3.0
    print(f"N={N}: potrf+trtri hand-written {t0:.3f} ms ({flops / t0 / 1e9:.1f} TFLOP/s), rocSOLVER "
          f"{t1:.3f} ms ({flops / t1 / 1e9:.1f} TFLOP/s), ratio {t1 / t0:.2f}; max|L-L'|/max|L| = "
          f"{np.max(np.abs(np.tril(L0) - np.tril(L1))) / np.max(np.abs(L0)):.1e}, "
          f"max|V-V'|/max|V| = {np.max(np.abs(np.tril(V0) - np.tril(V1))) / np.max(np.abs(V0)):.1e}", flush=True)
