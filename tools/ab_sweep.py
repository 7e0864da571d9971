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
variants = [("dma", 3, 32768, 0, 1, 0), ("sp_grid", 3, 32768, 0, 3, 0), ("sp_persist", 3, 32768, 0, 3, 9)]
res = {v[0]: [] for v in variants}
ref = None
for rnd in range(12):
    for name, tm, chunk, stg, xl, stag in variants:
        dev.set_option("sweep_persist", 1 if stag == 9 else 0)
        dev.set_option("sweep_stagger", 0)
        dev.set_option("sweep_extra_lds", 24576 if stag < 0 else 0)
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
dev.set_option("sweep_dma", 3)
for xl, ks in ((0, 0), (0, 1)):
    dev.set_option("sweep_extra_lds", xl)
    dev.set_option("sweep_persist", ks)
    dev.set_option("sweep_diag", 1)
    dev.read_diag(True)
    dev.timing_reset()
    dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
    dg = dev.read_diag(True).astype(float)
    ms, nl = dev.timing("sweep_gemm")
    dev.set_option("sweep_diag", 0)
    slots = 256 * (1 if xl else 2) * 4
    print(f"sp diag extra_lds={xl} {'persistent' if ks else 'grid'}: per wave-slab cycles: wait={dg[0] / dg[5]:.0f} barrier={dg[1] / dg[5]:.0f} "
          f"slab={dg[2] / dg[5]:.0f}; slabs/tile={dg[5] / dg[4]:.1f}; kernel {ms / nl:.3f} ms x{nl}; "
          f"wave-slot occupancy {dg[3] / 100e6 / (slots * ms * 1e-3):.4f}; "
          f"steady-state clock estimate {dg[2] / (dg[3] / 100e6) / 1e9:.3f} GHz (slab cycles / lifetime)")
dev.set_option("sweep_extra_lds", 0)
dev.set_option("sweep_persist", 0)
for name, v in res.items():
    print(f"{name}: TFLOP/s algorithmic median {np.median(v):.2f} min {min(v):.2f} max {max(v):.2f}  all " + " ".join(f"{x:.1f}" for x in v))
