#!/usr/bin/env python3
"""Sweep contraction: alternating k walk of the super-tiles of an XCD (option sweep_altwalk) against the plain map --
time per launch, agreement of sigma, and independence of the result from the chunking (bitwise)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N, d, M = 4096, 16, 262144
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N); Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
ref = {}
for alt in (0, 1, 0, 1, 0, 1):
    dev.set_option("sweep_altwalk", alt)
    out = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("sigma", "y"))
    dev.timing_reset()
    for _ in range(3): dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
    ms, n = dev.timing("sweep_gemm")
    ref.setdefault(alt, out)
    print(f"altwalk={alt}: sweep_gemm {ms / n:.3f} ms per launch of 32768, {M * 3 * (N * N + 2.0 * N) / (ms * 1e-3) / 1e12:.2f} TF; "
          f"same bits as first run of this mode: {np.array_equal(out['sigma'], ref[alt]['sigma'])}", flush=True)
s0, s1 = ref[0]["sigma"], ref[1]["sigma"]
print(f"sigma altwalk vs plain: max rel diff {np.max(np.abs(s1 - s0) / np.abs(s0)):.3e}; mean identical: {np.array_equal(ref[0]['y'], ref[1]['y'])}")
# chunking independence in altwalk mode (5120 -> 5 super-columns: the fallback map; 8192 -> 8; 16384 -> 16)
dev.set_option("sweep_altwalk", 1)
for chunk in (16384, 8192, 5120, 1024):
    dev.set_option("sweep_chunk", chunk)
    out = dev.sweep_logexp(Xc[:100000], 0.1, 0.0, 1e-2, want=("sigma",))
    print(f"chunk {chunk}: sigma bitwise equal to chunk 32768: {np.array_equal(out['sigma'], s1[:100000])}", flush=True)
# other training sizes (odd number of super-rows -> fallback map)
for N2 in (2048, 3000, 1000, 5000):
    X2 = rng.uniform(0, 1, (N2, d)); y2 = rng.standard_normal(N2)
    dev.set_option("sweep_chunk", 32768)
    dev.set_train(X2, y2, np.full(N2, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    assert dev.factorize() == 0
    res = []
    for alt in (0, 1):
        dev.set_option("sweep_altwalk", alt)
        out = dev.sweep_logexp(Xc[:65536], 0.1, 0.0, 1e-2, want=("sigma",))
        dev.timing_reset()
        for _ in range(3): dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=65536, want=())
        ms, n = dev.timing("sweep_gemm")
        res.append((out["sigma"], ms / n))
    print(f"N={N2}: plain {res[0][1]:.3f} ms, altwalk {res[1][1]:.3f} ms per launch; max rel diff {np.max(np.abs(res[1][0] - res[0][0]) / np.abs(res[0][0])):.2e}", flush=True)
