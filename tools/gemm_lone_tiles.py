import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
dev = _lib.Device(0)
rng = np.random.default_rng(0)
for (M, N, K) in ((2048, 2048, 4096), (2048, 4096, 4096), (4096, 4096, 4096), (2048, 2048, 1024)):
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N))
    for name, at, bt in (("NN", 0, 0), ("NT", 0, 1), ("TN", 1, 0)):
        Ain = np.ascontiguousarray(A.T) if at else A
        Bin = np.ascontiguousarray(B.T) if bt else B
        for dma in (1, 0):
            dev.set_option("gemm_dma", dma)
            for _ in range(2):
                dev.timing_reset()
                dev.debug_gemm(Ain, Bin, None, M, N, K, at, bt)
                ms, _ = dev.timing("debug_gemm")
            print(f"{M}x{N}x{K} {name} dma={dma}: {ms:.3f} ms, {2.0*M*N*K/ms/1e9:.1f} TF, tiles {M*N//16384}, "
                  f"cycles/slab if one round: {ms*1e-3*2.4e9/(K/16)/max(1,(M*N//16384+511)//512 if M*N//16384>256 else 1):.0f}", flush=True)
dev.set_option("gemm_dma", 1)
