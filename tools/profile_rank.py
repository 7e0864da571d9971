#!/usr/bin/env python3
"""cProfile of NORA.multi_add at the bench shape (N=4096, d=16): where the host time of the ranking goes."""
import cProfile
import os
import pstats
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gpry_amd.gp_acquisition import NORA  # noqa: E402

N, d = 4096, 16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
bounds, X, y, Xc, truth = bench.synthetic(N, d, M)
gpr = bench.make_gpr(bounds)
gpr.append_to_data(X, y, fit_gpr="simple")
acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
rng = np.random.default_rng(2)
for _ in range(3):
    acq.multi_add(gpr, n_points=d, rng=rng)
t0 = time.perf_counter()
for _ in range(5):
    acq.multi_add(gpr, n_points=d, rng=rng)
print(f"multi_add: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call; stats {acq.stats}")
# the bench cycle: the proposals are appended, the hyper-parameters refitted, then the next multi_add
n_base = N
X_new, _, _ = acq.multi_add(gpr, n_points=d, rng=rng)
for _ in range(2):
    bench.rewind(gpr, n_base)
    gpr.append_to_data(X_new, truth(X_new), fit_gpr="simple")
    X_new, _, _ = acq.multi_add(gpr, n_points=d, rng=rng)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    bench.rewind(gpr, n_base)
    gpr.append_to_data(X_new, truth(X_new), fit_gpr="simple")
    X_new, _, _ = acq.multi_add(gpr, n_points=d, rng=rng)
pr.disable()
print("stats", acq.stats)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
