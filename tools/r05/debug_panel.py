#!/usr/bin/env python3
"""Diagnostic: small factorisations through the stamped build, status words printed (python3 tools/r05/debug_panel.py N ...)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GPRY_HIP_LIB"] = os.path.join(ROOT, "tools", "r05", "libgpry_hip_stamps.so")
from gpry_amd import _lib
print("creating device", flush=True)
dev = _lib.Device(0)
print("device created", flush=True)
dev.set_option("factor_pipeline", 0)
lib = _lib.load_library()
lib.gpry_debug_read_info.restype = C.c_int
lib.gpry_debug_read_info.argtypes = [C.c_void_p, C.c_void_p]
lib.gpry_debug_progress_buffer.restype = C.POINTER(C.c_int)
prog = lib.gpry_debug_progress_buffer()
import threading
def watchdog():
    time.sleep(8)
    a = np.ctypeslib.as_array(prog, shape=(16, 4, 8))
    print("HANG? progress codes [step][workgroup][wave]:", flush=True)
    for st in range(4):
        print("  step", st, a[st].tolist(), flush=True)
    os._exit(3)
threading.Thread(target=watchdog, daemon=True).start()
for N in [int(a) for a in (sys.argv[1:] or ["100", "200", "1000"])]:
    d = 3
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    print(f"N={N}: train set, factorising", flush=True)
    t0 = time.time()
    try:
        rc = dev.factorize()
    except Exception as e:
        rc = repr(e)
    dt = time.time() - t0
    inf = np.zeros(4, dtype=np.int32)
    lib.gpry_debug_read_info(dev._h, inf.ctypes.data_as(C.c_void_p))
    print(f"N={N}: factorize -> {rc} in {dt:.3f} s; dinfo = {[hex(int(v)) for v in inf]}", flush=True)
    if rc == 0:
        L = np.tril(dev.get_factor()[0])
        K = L @ L.T
        for ov in (0,):
            dev.set_option("chol_overlap", ov)
            dev.factorize()
            L2 = np.tril(dev.get_factor()[0])
            print(f"   overlap=1 vs overlap={ov}: bit-identical {np.array_equal(L, L2)}; |LL^T| ok {np.isfinite(K).all()}", flush=True)
            dev.set_option("chol_overlap", 1)
