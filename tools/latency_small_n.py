#!/usr/bin/env python3
"""Small-N regime (where real GPry runs spend most iterations): wall-clock latency per call on the GPU
next to the CPU port (oracle/gpry_oracle.py, numpy/scipy on the host cores) for N = 64 / 256 / 1024:
LML+gradient, predict (mean; mean+std), one NORA.multi_add (M = 1000 d uniform candidates, n_points = d).
Prints a markdown table (kept as profiles/r02_latency_small_n.md)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gpry_oracle as orc            # CPU baseline beside the GPU numbers (test/bench infrastructure)
from gpry_amd.gpr import GaussianProcessRegressor
from gpry_amd.gp_acquisition import NORA
from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
from gpry_amd.kernels import clone


def timed(fn, reps, warm=2):
    """Median over 5 batches after at least 30 ms of warm-up (GPU clocks ramp up, BLAS pools settle)."""
    t0 = time.perf_counter()
    n = 0
    while n < warm or (time.perf_counter() - t0 < 0.03 and n < 1000):
        fn()
        n += 1
    per = max(1, reps // 5)
    batches = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(per):
            fn()
        batches.append((time.perf_counter() - t0) / per * 1e6)
    return sorted(batches)[2]


rows = []
for N, d in ((64, 2), (256, 4), (1024, 8)):
    M = 1000 * d
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=3)
    theta = np.log(np.array([4.0] + [0.3] * d))
    ref = orc.OracleGPR(bounds, kernel_id=orc.MATERN52)
    ref.theta = theta
    ref.fitted = True
    ref.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    gpr = GaussianProcessRegressor(kernel={"Matern": {"nu": 2.5}}, bounds=bounds, preprocessing_X=Normalize_bounds(bounds),
                                   preprocessing_y=Normalize_y(), account_for_inf=None)
    k = clone(gpr.kernel); k.theta = theta
    gpr.kernel_, gpr._fitted = k, True
    gpr.append_to_data(X, y, fit_gpr=False)
    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    x1, x64 = Xc[:1], Xc[:64]
    cases = [
        ("LML + gradient", lambda: gpr.log_marginal_likelihood(theta, eval_gradient=True),
         lambda: orc.log_marginal_likelihood(ref.X_train_, ref.y_train_, ref.alpha, theta, orc.MATERN52, eval_gradient=True), 50, 5),
        ("predict mean, 1 point", lambda: gpr.predict(x1), lambda: ref.predict(x1), 200, 50),
        ("predict mean + std, 1 point", lambda: gpr.predict(x1, return_std=True), lambda: ref.predict(x1, return_std=True), 200, 50),
        ("predict mean + std, 64 points", lambda: gpr.predict(x64, return_std=True), lambda: ref.predict(x64, return_std=True), 100, 20),
        (f"multi_add, M={M}, n_points={d}", lambda: acq.multi_add(gpr, n_points=d, rng=np.random.default_rng(2)),
         lambda: orc.nora_multi_add(ref, Xc, d), 10, 2),
    ]
    # all GPU cases first: a multi-threaded BLAS call leaves its 256 worker threads spinning for a while,
    # which slows the Python thread that drives the GPU (a mixed order showed 2 ms for a 0.3-ms call)
    tgs = [timed(g, rg) for name, g, c, rg, rc in cases]
    for (name, g, c, rg, rc), tg in zip(cases, tgs):
        rows.append((N, d, name, tg, timed(c, rc, warm=1)))
    time.sleep(0.5)

print("| N | d | call | GPU (us) | CPU port (us) | CPU / GPU |")
print("|---|---|---|---|---|---|")
for N, d, name, tg, tc in rows:
    print(f"| {N} | {d} | {name} | {tg:.0f} | {tc:.0f} | {tc / tg:.1f} |")
try:
    from threadpoolctl import threadpool_info
    thr = max(p.get("num_threads", 1) for p in threadpool_info())
except Exception:
    thr = os.cpu_count()
print(f"\nHost: {os.cpu_count()} logical CPUs, {thr} BLAS threads.  Wall clock per call (median of 5 batches), Python overhead of both sides included.")
