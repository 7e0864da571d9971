"""One-point mean predict: resident kernel (predict_serve=1) vs one launch per call (0), at the C ABI (ctypes) and
through the mirror class.  usage: python tools/latency_serve.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpry_amd import _lib
from gpry_amd.gpr import GaussianProcessRegressor
from gpry_amd.kernels import clone
from gpry_amd.preprocessing import Normalize_bounds, Normalize_y

print("| N | d | path | us per call: ctypes Device.predict | us per call: GaussianProcessRegressor.predict |")
print("|---|---|---|---|---|")
for N, d in ((64, 2), (256, 4), (1024, 8), (4096, 16)):
    rng = np.random.default_rng(0)
    bounds = np.array([[-5.0, 5.0]] * d)
    X = rng.uniform(-5, 5, size=(N, d)); y = -0.5 * (X ** 2).sum(1)
    gpr = GaussianProcessRegressor(kernel={"Matern": {"nu": 2.5}}, bounds=bounds, preprocessing_X=Normalize_bounds(bounds),
                                   preprocessing_y=Normalize_y(), account_for_inf=None, verbose=0)
    k = clone(gpr.kernel); k.theta = np.log(np.array([4.0] + [0.3] * d)); gpr.kernel_ = k; gpr._fitted = True
    gpr.append_to_data(X, y, fit_gpr=False)
    Xq = rng.uniform(-5, 5, size=(4000, d))
    gpr.predict(Xq[:1])
    dev = gpr.device
    for serve in (1, 0):
        dev.set_option("predict_serve", serve)
        dev.set_option("serve_idle_us", 5000)
        for x in Xq[:200]: dev.predict(x[None, :])
        t0 = time.perf_counter()
        for x in Xq: dev.predict(x[None, :])
        t_c = (time.perf_counter() - t0) / len(Xq) * 1e6
        t0 = time.perf_counter()
        for x in Xq: gpr.predict(np.atleast_2d(x), return_std=False, validate=False)
        t_p = (time.perf_counter() - t0) / len(Xq) * 1e6
        print(f"| {N} | {d} | {'resident kernel' if serve else 'one launch per call'} | {t_c:.1f} | {t_p:.1f} |", flush=True)
    print(f"| {N} | {d} | (generations, requests) | {dev.serve_stats()} | |", flush=True)
