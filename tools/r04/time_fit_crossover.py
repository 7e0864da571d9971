#!/usr/bin/env python3
"""Where does the side-by-side fit on the batched chain stop paying against the thread farm of three contexts?  Full fits
(10 + 2 d restarts) at N = 2048 ... 4096."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for N, d in [(int(a), 8) for a in (sys.argv[1:] or ["2048", "2560", "3072", "3584", "4096"])]:
    bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
    res = {}
    for mode, env in (("3 contexts", ("3", "0", "1")), ("side by side", ("3", "1", "1")), ("x2 groups", ("3", "1", "2")), ("x3", ("3", "1", "3")), ("x4", ("4", "1", "4")), ("x6", ("6", "1", "6"))):
        os.environ["GPRY_HIP_FIT_CONTEXTS"], os.environ["GPRY_HIP_FIT_LOCKSTEP"], os.environ["GPRY_HIP_FIT_BATCH_CONTEXTS"] = env
        best = None
        gpr = bench.make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)       # ONE model, as in a run: contexts and arenas persist
        gpr.append_to_data(X, y, fit_gpr=False)
        for rep in range(3):
            gpr.set_random_state(3)
            e0 = gpr.n_eval_loglike
            t0 = time.perf_counter()
            gpr.fit_gpr_hyperparameters(start_from_current=False)
            dt = time.perf_counter() - t0
            if rep:
                best = dt if best is None else min(best, dt)
            nev = gpr.n_eval_loglike - e0
        res[mode] = (best, nev, gpr.kernel_.theta.copy())
    same = np.array_equal(res["side by side"][2], res["3 contexts"][2])
    print(f"N={N} d={d} ({10 + 2 * d} restarts, {res['3 contexts'][1]} evaluations): " +
          " | ".join(f"{k} {v[0] * 1e3:.0f} ms" for k, v in res.items()) + f" (same optimum: {same})", flush=True)
