#!/usr/bin/env python3
"""TFLOP/s of the generic MFMA GEMM engine in its four layouts and triangular modes (n^3)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
A = rng.standard_normal((n, n)); B = rng.standard_normal((n, n)); C0 = np.zeros((n, n))
dev = _lib.Device(0)
cases = [("NN full", 0, 0, 0, 0, 0, 2.0), ("NT full", 0, 1, 0, 0, 0, 2.0), ("TN full", 1, 0, 0, 0, 0, 2.0),
         ("TT full", 1, 1, 0, 0, 0, 2.0), ("NN A lower (trtri V)", 0, 0, 0, 1, 0, 1.0),
         ("NN B lower (trtri T)", 0, 0, 0, 2, 0, 1.0), ("TN A^T B both lower, lower tiles (lauum)", 1, 0, 0, 3, 1, 1.0 / 3.0),
         ("NT C-=AB^T lower tiles (syrk shape, K=n)", 0, 1, 2, 0, 1, 1.0)]
ref = {}
for name, at, bt, epi, kmode, lo, frac in cases:
    for nsplit in ((1,) if epi == 2 else (1, 0x100)):       # 0x100: marker for "XCD-aware super-tile map"
        for _ in range(2):
            dev.timing_reset()
            C = dev.debug_gemm(A, B, C0.copy(), n, n, n, a_trans=bool(at), b_trans=bool(bt), epi=epi, kmode=kmode,
                               lower_only=lo, tile_map=(1 | (3 << 4)) if nsplit == 0x100 else (nsplit << 8))
            ms, cnt = dev.timing("debug_gemm")
        if nsplit == 1:
            ref[name] = C
        err = float(np.max(np.abs(C - ref[name]))) / max(1e-300, float(np.max(np.abs(ref[name]))))
        print(f"{name:45s} {'xcd-map' if nsplit == 0x100 else 'row-major'}: {ms:8.3f} ms  {frac * n ** 3 / (ms * 1e-3) / 1e12:6.1f} TFLOP/s useful; "
              f"vs unsplit {err:.1e}")
