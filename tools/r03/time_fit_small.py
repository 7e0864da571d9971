#!/usr/bin/env python3
"""A full multi-restart hyper-parameter fit at small N (what GPry does every few iterations early in a run): the restarts one
after another in one context, shared by three contexts / host threads, and stepped side by side with batched objective calls."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for N, d in ((40, 2), (64, 4), (128, 8), (128, 16)):
    bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
    res = {}
    for mode, env in (("one after another", {"GPRY_HIP_FIT_CONTEXTS": "1", "GPRY_HIP_FIT_LOCKSTEP": "0"}),
                      ("three contexts", {"GPRY_HIP_FIT_CONTEXTS": "3", "GPRY_HIP_FIT_LOCKSTEP": "0"}),
                      ("side by side", {"GPRY_HIP_FIT_CONTEXTS": "1", "GPRY_HIP_FIT_LOCKSTEP": "1"})):
        os.environ.update(env)
        best = None
        for rep in range(3):
            gpr = bench.make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)
            gpr.append_to_data(X[:4], y[:4], fit_gpr=False)          # context, allocations
            t0 = time.perf_counter()
            gpr.append_to_data(X[4:], y[4:], fit_gpr=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[mode] = (best, gpr.n_eval_loglike, gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_)
    ref = res["one after another"]
    print(f"N={N} d={d} ({10 + 2 * d} restarts, {ref[1]} evaluations): " + ", ".join(
        f"{m} {v[0] * 1e3:.1f} ms" + ("" if m == "one after another" else f" (same optimum: {np.array_equal(v[2], ref[2]) and v[3] == ref[3]})")
        for m, v in res.items()), flush=True)
