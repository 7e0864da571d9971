#!/usr/bin/env python3
"""Small driver for rocprofv3 counter passes: factor once, then sweep M candidates."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = int(sys.argv[3]) if len(sys.argv) > 3 else 131072
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 32768
dma = int(sys.argv[5]) if len(sys.argv) > 5 else 1       # 0: the register-staged comparator ("gemm_dma" = 0)
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d))
y = rng.standard_normal(N)
Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_option("sweep_chunk", chunk)
dev.set_option("gemm_dma", dma)
dev.set_train(X, y, np.full(N, 1e-4))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=())
dev.timing_reset()
for _ in range(2):
    dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
for k in ("cross_build", "sweep_gemm", "sweep_finish"):
    ms, n = dev.timing(k)
    print(f"{k}: {ms / max(n, 1):.3f} ms avg over {n}")
ms, n = dev.timing("sweep_gemm")
print(f"sweep_gemm algorithmic TFLOP/s: {2 * M * (N * N + 2.0 * N) / (ms * 1e-3) / 1e12:.2f}")
