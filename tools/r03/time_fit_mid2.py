#!/usr/bin/env python3
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
for N, d in ((400, 6), (800, 8)):
    bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
    res = {}
    for n_ctx in ("1", "3", "4", "6", "8"):
        os.environ["GPRY_HIP_FIT_CONTEXTS"] = n_ctx
        best = None
        for rep in range(2):
            gpr = bench.make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)
            gpr.append_to_data(X[:4], y[:4], fit_gpr=False)
            t0 = time.perf_counter()
            gpr.append_to_data(X[4:], y[4:], fit_gpr=True)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[n_ctx] = best
    print(f"N={N} d={d}: " + ", ".join(f"{k} ctx {v * 1e3:.0f} ms" for k, v in res.items()), flush=True)
