#!/usr/bin/env python3
"""Stress at a few hundred training points (general factor chain, small-tile GEMMs, 128 x 128 diagonal stage): the fit
with restarts shared by three device contexts must reproduce the sequential one bit for bit, over and over."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_host_mirror_gpu import make_gpr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
d = 4
rng = np.random.default_rng(5)
bounds = np.array([[0.0, 1.0]] * d)
X = rng.uniform(size=(N, d)); y = -20.0 * ((X - 0.4) ** 2).sum(1) + 0.01 * rng.standard_normal(N)
Xc = rng.uniform(size=(40, d))
ref, bad = None, 0
for it in range(n):
    os.environ["GPRY_HIP_FIT_CONTEXTS"] = "3" if it else "1"
    gpr = make_gpr(bounds, 3, n_restarts_optimizer=6, random_state=11)
    gpr.append_to_data(X, y, fit_gpr=True)
    m, s = gpr.predict(Xc, return_std=True)
    out = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike, m, s)
    if ref is None:
        ref = out
        continue
    if not (np.array_equal(out[0], ref[0]) and out[1] == ref[1] and out[2] == ref[2] and np.array_equal(out[3], ref[3]) and np.array_equal(out[4], ref[4])):
        bad += 1
        print(f"iteration {it}: evals {out[2]} vs {ref[2]}, |dtheta| {np.max(np.abs(out[0] - ref[0])):.2e}, dlml {out[1] - ref[1]:.2e}", flush=True)
print(f"N={N}: {bad} mismatches in {n - 1} concurrent fits of {ref[2]} evaluations each")
