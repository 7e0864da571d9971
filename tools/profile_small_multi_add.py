#!/usr/bin/env python3
"""cProfile of NORA.multi_add at small N (launch-bound regime): which device calls make up the ~0.8 ms."""
import cProfile
import os
import pstats
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gpry_amd.gp_acquisition import NORA  # noqa: E402

N, d, M = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 4, 4000)))
bounds, X, y, Xc, truth = bench.synthetic(N, d, M)
gpr = bench.make_gpr(bounds)
gpr.append_to_data(X, y, fit_gpr="simple")
acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
rng = np.random.default_rng(2)
for _ in range(5):
    acq.multi_add(gpr, n_points=d, rng=rng)
t0 = time.perf_counter()
for _ in range(50):
    acq.multi_add(gpr, n_points=d, rng=rng)
print(f"multi_add N={N} d={d} M={M}: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us per call; stats {acq.stats}")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    acq.multi_add(gpr, n_points=d, rng=rng)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
