#!/usr/bin/env python3
"""BASELINE config 5 (N=8192, d=20, 32 restarts over 8 GPUs): time of ONE GPU's share of the restart
farm (4 restarts, the first from the current theta) through gpry_amd.parallel.fit_gpr_parallel."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gpry_amd.gpr import GaussianProcessRegressor  # noqa: E402
from gpry_amd.parallel import fit_gpr_parallel, split_number_for_parallel_processes  # noqa: E402
from gpry_amd.preprocessing import Normalize_bounds, Normalize_y  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 20
total, world = 32, 8
bounds, X, y, _, _ = bench.synthetic(N, d, 16)
gpr = GaussianProcessRegressor(kernel={"Matern": {"nu": 2.5}}, bounds=bounds, noise_level=1e-2,
                               preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
                               account_for_inf=None, random_state=3, n_restarts_optimizer=total, verbose=0)
gpr.append_to_data(X[:-d], y[:-d], fit_gpr="simple")          # a fitted model to start from
share = int(split_number_for_parallel_processes(total, world)[0])
import copy  # noqa: E402
for n_ctx in ("1", "2", "3"):       # device contexts (host threads) sharing the restarts of this rank
    os.environ["GPRY_HIP_FIT_CONTEXTS"] = n_ctx
    g2 = copy.deepcopy(gpr)
    g2.predict(X[:2])                                            # factor in place, as in a running loop
    g2.device.timing_reset()
    n0 = g2.n_eval_loglike
    t0 = time.perf_counter()
    lml, best, _ = fit_gpr_parallel(g2, X[-d:], y[-d:], comm=None, fit="full", n_restarts=share)
    dt = time.perf_counter() - t0
    ne = g2.n_eval_loglike - n0
    print(f"N={N} d={d}, {n_ctx} context(s): {share} restarts (1/{world} of {total}) in {dt:.2f} s, {ne} LML+grad "
          f"evaluations, {dt / ne * 1e3:.1f} ms each (aggregate); best lml {lml:.6f}")
    if n_ctx == "1":
        for k in ("potrf", "trtri", "lauum", "lml_traces", "kernel_build"):
            ms, n = g2.device.timing(k)
            print(f"  {k}: {ms / max(n, 1):.2f} ms avg over {n}")
