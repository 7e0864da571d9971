#!/usr/bin/env python3
"""cProfile of BatchOptimizer.multi_add (side by side) at N = 256 / 1024."""
import os, sys, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gpry_amd.gp_acquisition import BatchOptimizer
N, d = int(sys.argv[1]), int(sys.argv[2])
bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
gpr = bench.make_gpr(bounds)
gpr.append_to_data(X, y, fit_gpr="simple")
acq = BatchOptimizer(bounds, n_restarts_optimizer=5 * d, n_repeats_propose=10, verbose=0)
rng = np.random.default_rng(1)
acq.multi_add(gpr, n_points=2, rng=rng)
pr = cProfile.Profile(); pr.enable()
acq.multi_add(gpr, n_points=2, rng=rng)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
