#!/usr/bin/env python3
"""BatchOptimizer.multi_add (GPry's acquisition engine when no nested sampler is installed) through the mirror classes:
wall clock per call and per posterior evaluation, with the call counts of the device entry points."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gpry_amd.gp_acquisition import BatchOptimizer
for N, d in ((64, 2), (256, 4), (1024, 8)):
    bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
    gpr = bench.make_gpr(bounds)
    gpr.append_to_data(X, y, fit_gpr="simple")
    for lock in (False, "auto"):
        acq = BatchOptimizer(bounds, n_restarts_optimizer=5 * d, n_repeats_propose=10, verbose=0, lockstep=lock)
        rng = np.random.default_rng(1)
        acq.multi_add(gpr, n_points=2, rng=rng)
        e0 = gpr.n_eval
        t0 = time.perf_counter()
        Xn, yl, av = acq.multi_add(gpr, n_points=2, rng=rng)
        dt = time.perf_counter() - t0
        ne = gpr.n_eval - e0
        print(f"N={N} d={d} {'side by side' if acq.stats['side_by_side'] else 'one after another'}: multi_add(n_points=2, {5 * d} restarts) {dt * 1e3:.1f} ms, "
              f"{ne} posterior evaluations, {dt / max(ne, 1) * 1e6:.1f} us each; first proposal {np.round(Xn[0], 5)}", flush=True)
    if N == -1:
        pr = cProfile.Profile(); pr.enable()
        acq.multi_add(gpr, n_points=2, rng=rng)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
