#!/usr/bin/env python3
"""gpry_lml_batch above N = 128 (ONE chain of launches for all thetas) against the same thetas one after another: time per
call for B = 2 ... 64 at N = 256 ... 2048, then full multi-restart fits (10 + 2 d restarts, scipy L-BFGS-B) at the shapes of
tools/r03/time_fit_mid.py: restarts one after another, the thread farm of 3 contexts (round 3's default above 128 points)
and stepped side by side on the batched chain (the default now).  Writes one line per measurement."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gpry_amd import _lib

def best_of(f, reps=5):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)

sizes = [(256, 4), (400, 6), (800, 8), (1024, 8), (1600, 8), (2048, 16), (4096, 16), (8192, 20)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sizes = [(400, 6), (1024, 8)]
dv = _lib.Device(0)
rng = np.random.default_rng(0)
print("# time per call (ms): B thetas in ONE chain (gpry_lml_batch) | one after another (B x gpry_lml) | ratio; LML + gradient")
for N, d in sizes:
    X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1); y = (y - y.mean()) / y.std()
    dv.set_train(X, y, np.full(N, 1e-4)); kid = 0 if (N, d) == (1024, 8) else 3
    base = np.log(np.array([2.0] + [0.5] * d)); dv.set_theta(kid, base)
    t1 = best_of(lambda: dv.lml(base, True))
    row = [f"N={N} d={d} kid={kid}: single {t1 * 1e3:.3f}"]
    for B in ((2, 4, 8, 16, 32, 64) if N <= 2048 else (2, 4, 8, 16) if N <= 4096 else (2, 4, 8)):
        th = base + rng.uniform(-0.3, 0.3, (B, d + 1))
        tb = best_of(lambda: dv.lml_batch(th, True), reps=3)
        row.append(f"B={B} {tb * 1e3:.2f} | {B * t1 * 1e3:.2f} | x{B * t1 / tb:.1f}")
    print("; ".join(row), flush=True)
dv.close()

print("# full fit (ms, best of 2; evaluations): one after another | 3 contexts | side by side (same optimum as sequential?)")
for N, d in ((200, 4), (400, 6), (800, 8), (1600, 8), (4096, 16)) if not (len(sys.argv) > 1 and sys.argv[1] == "quick") else ((400, 6),):
    bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
    res = {}
    for mode, env in (("sequential", ("1", "0")), ("3 contexts", ("3", "0")), ("side by side", ("1", "1"))):
        os.environ["GPRY_HIP_FIT_CONTEXTS"], os.environ["GPRY_HIP_FIT_LOCKSTEP"] = env
        best = None
        for rep in range(2):
            gpr = bench.make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)
            gpr.append_to_data(X[:4], y[:4], fit_gpr=False)
            t0 = time.perf_counter()
            gpr.append_to_data(X[4:], y[4:], fit_gpr=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[mode] = (best, gpr.n_eval_loglike, gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, getattr(gpr, "fit_stats", None))
    same = np.array_equal(res["side by side"][2], res["sequential"][2]) and res["side by side"][3] == res["sequential"][3]
    rounds = max(res["side by side"][4]["evals_per_run"]) if res["side by side"][4] and "evals_per_run" in res["side by side"][4] else -1
    print(f"N={N} d={d} ({10 + 2 * d} restarts, {res['sequential'][1]} evaluations, {rounds} rounds): " +
          " | ".join(f"{k} {v[0] * 1e3:.0f} ms" for k, v in res.items()) + f" (same optimum: {same})", flush=True)
