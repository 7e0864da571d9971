#!/usr/bin/env python3
"""cProfile of the refit half of the bench cycle (append 16 points + fit_gpr='simple') at N=4096."""
import cProfile
import os
import pstats
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

N, d = 4096, 16
bounds, X, y, Xc, truth = bench.synthetic(N - d, d, 1000)
gpr = bench.make_gpr(bounds)
gpr.append_to_data(X, y, fit_gpr="simple")
rng = np.random.default_rng(5)
Xn = np.clip(rng.standard_normal((d, d)), -5, 5)
for _ in range(3):
    bench.rewind(gpr, N - d)
    gpr.append_to_data(Xn, truth(Xn), fit_gpr="simple")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    bench.rewind(gpr, N - d)
    gpr.append_to_data(Xn, truth(Xn), fit_gpr="simple")
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
