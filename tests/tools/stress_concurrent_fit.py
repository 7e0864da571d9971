#!/usr/bin/env python3
"""Stress: the fit with restarts shared by three device contexts must reproduce the sequential one bit for bit,
over and over (races in the result hand-over show up as different evaluation counts or hyper-parameters)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import load_golden  # noqa: E402
from test_host_mirror_gpu import make_gpr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = load_golden("fit")
p = "f6_k3_"
X, y = g[p + "X"], g[p + "y"]
ref = None
bad = 0
for it in range(n):
    os.environ["GPRY_HIP_FIT_CONTEXTS"] = "3" if it else "1"
    gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=6, random_state=11)
    gpr.append_to_data(X[:60], y[:60], fit_gpr=True)
    out = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike, gpr.predict(g[p + "Xc"]))
    if ref is None:
        ref = out
        continue
    if not (np.array_equal(out[0], ref[0]) and out[1] == ref[1] and out[2] == ref[2] and np.array_equal(out[3], ref[3])):
        bad += 1
        print(f"iteration {it}: evals {out[2]} vs {ref[2]}, |dtheta| {np.max(np.abs(out[0] - ref[0])):.2e}, "
              f"dlml {out[1] - ref[1]:.2e}", flush=True)
print(f"{bad} mismatches in {n - 1} concurrent fits (GPRY_HIP_OPTIONS={os.environ.get('GPRY_HIP_OPTIONS', '')})")
sys.exit(1 if bad else 0)
