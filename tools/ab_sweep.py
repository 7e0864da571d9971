#!/usr/bin/env python3
"""Interleaved A/B of sweep-GEMM launch options in ONE process (guide rule 24)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d, M = 4096, 16, 262144
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N); Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
variants = [("dma", 3, 65536, 0, 1, 0), ("dma_setprio", 3, 65536, 0, 1, 7)]
res = {v[0]: [] for v in variants}
ref = None
for rnd in range(3):
    for name, tm, chunk, stg, xl, stag in variants:
        dev.set_option("sweep_stagger", stag)
        dev.set_option("sweep_dma", xl)
        dev.set_option("sweep_kskew", stg)
        dev.set_option("sweep_tilemap", tm)
        dev.set_option("sweep_chunk", chunk)
        out = dev.sweep_logexp(Xc if rnd == 0 else None, 0.1, 0.0, 1e-2, M=M, want=("acq",))
        if ref is None:
            ref = out["acq"]
        assert stg == 77 or np.allclose(out["acq"], ref, rtol=1e-9, atol=1e-9, equal_nan=True), name
        dev.timing_reset()
        dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
        ms, n = dev.timing("sweep_gemm")
        res[name].append(M * (N * N + 2.0 * N) / (ms * 1e-3) / 1e12)
dev.set_option("sweep_dma", 0)
for xl in (0,):
    dev.set_option("sweep_diag", 1)
    dev.read_diag(True)
    dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
    dg = dev.read_diag(True).astype(float)
    dev.set_option("sweep_diag", 0)
    names = ["issue_loads", "mfma_block", "vmcnt_wait", "lds_store", "barrier"]
    print(f"diag extra_lds={xl}: per wave-slab cycles: " + ", ".join(f"{n}={dg[i] / dg[5]:.0f}" for i, n in enumerate(names)))
for name, v in res.items():
    print(f"{name}: TFLOP/s algorithmic median {np.median(v):.2f} min {min(v):.2f} max {max(v):.2f}")
