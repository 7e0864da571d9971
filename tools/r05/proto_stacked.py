#!/usr/bin/env python3
"""PROTOTYPE: the inverse factor as extra rows of the panel chain -- [K; I] (2 Np x Np) through one rectangular segment of the
overlap schedule (gpry_proto_potrf_stacked, tools/r05/build_proto.sh).  Checks L and U = L^-T against numpy and times the
stacked chain against potrf alone and against factorize (potrf + V = L^-1) of the product library:
    GPRY_HIP_LIB=tools/r05/libgpry_hip_proto.so python3 tools/r05/proto_stacked.py 256 512 1024 2048 ..."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
lib = _lib.load_library()
lib.gpry_proto_potrf_stacked.restype = ctypes.c_int
lib.gpry_proto_potrf_stacked.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
dev = _lib.Device(0)
for N in [int(a) for a in (sys.argv[1:] or ["256", "512", "1024", "2048"])]:
    d = 8
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    r2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / 0.3 ** 2 if N <= 2048 else None
    if r2 is None:
        G = X @ X.T; sq = np.diag(G); r2 = (sq[:, None] + sq[None, :] - 2 * G) / 0.3 ** 2
    r = np.sqrt(np.maximum(r2, 0) * 5.0)
    K = 4.0 * (1 + r + r * r / 3) * np.exp(-r) + 1e-4 * np.eye(N)
    K = np.ascontiguousarray(K)
    L = np.zeros((N, N)); U = np.zeros((N, N))
    ms_s, ms_q = ctypes.c_double(0), ctypes.c_double(0)
    rc = lib.gpry_proto_potrf_stacked(dev._h, K.ctypes.data, N, L.ctypes.data, U.ctypes.data, 5, ctypes.byref(ms_s), ctypes.byref(ms_q))
    assert rc == 0, rc
    Lr = np.linalg.cholesky(K)
    Ur = np.linalg.inv(Lr).T
    eL = np.abs(np.tril(L) - Lr).max() / np.abs(Lr).max()
    eU = np.abs(np.triu(U) - Ur).max() / np.abs(Ur).max()
    low = np.abs(np.tril(U, -1)).max() / np.abs(Ur).max()         # the strictly lower part of U: zero in exact arithmetic
    # the product library's factorize (potrf + V = L^-1) on the same matrix size
    dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    dev.set_option("factor_pipeline", 0)
    assert dev.factorize() == 0
    dev.timing_reset()
    for _ in range(5):
        dev.factorize()
    tp = dev.timing("potrf"); tt = dev.timing("trtri")
    dev.set_option("timing", 0)
    print(f"N={N}: stacked chain {ms_s.value * 1e3:.0f} us | square chain {ms_q.value * 1e3:.0f} us | product: potrf {tp[0] / max(tp[1], 1) * 1e3:.0f} us + "
          f"V = L^-1 {tt[0] / max(tt[1], 1) * 1e3:.0f} us || L err {eL:.1e}, U = L^-T err {eU:.1e}, below the diagonal of U {low:.1e}")
