#!/usr/bin/env python3
"""Where the wall time of a side-by-side fit goes on the host (cProfile): python3 tools/r04/profile_fit_host.py N d"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
N, d = int(sys.argv[1]), int(sys.argv[2])
bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
gpr = bench.make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)
gpr.append_to_data(X, y, fit_gpr=False)
gpr.set_random_state(3); gpr.fit_gpr_hyperparameters(start_from_current=False)
gpr.set_random_state(3)
t0 = time.perf_counter(); gpr.fit_gpr_hyperparameters(start_from_current=False); print(f"N={N} d={d}: fit {1e3 * (time.perf_counter() - t0):.1f} ms, rounds {max(gpr.fit_stats['evals_per_run'])}, evaluations {sum(gpr.fit_stats['evals_per_run'])}")
gpr.set_random_state(3)
pr = cProfile.Profile(); pr.enable(); gpr.fit_gpr_hyperparameters(start_from_current=False); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
