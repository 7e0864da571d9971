#!/usr/bin/env python3
"""Factor chain with the DMA-pipelined GEMM (gemm_dma=1) vs the register-staged engine (0)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = _lib.Device(0)
for N, d in ((1024, 8), (2048, 12), (4096, 16), (8192, 20)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d))
    y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d))
    dev.set_theta(3, theta)
    res = {}
    for dma in (0, 1, 0, 1):
        dev.set_option("gemm_dma", dma)
        dev.lml(theta, True)
        dev.timing_reset()
        for _ in range(reps):
            out = dev.lml(theta, True)
        t = {k: dev.timing(k)[0] / reps for k in ("potrf", "trtri", "lauum", "lml_traces", "kernel_build")}
        res[dma] = (t, out)
        print(f"N={N} gemm_dma={dma}: potrf {t['potrf']:.3f} trtri {t['trtri']:.3f} lauum {t['lauum']:.3f} ms; "
              f"sum {sum(t.values()):.3f} ms; lml {out[0]:.12g}", flush=True)
    g0, g1 = res[0][1][1], res[1][1][1]
    print(f"   lml equal: {res[0][1][0] == res[1][1][0]}; max |grad diff| / max |grad| = "
          f"{np.max(np.abs(g0 - g1)) / np.max(np.abs(g0)):.1e}")
