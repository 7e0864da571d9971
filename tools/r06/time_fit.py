#!/usr/bin/env python3
"""Full multi-restart fit (10 + 2 d restarts, scipy L-BFGS-B) of bench.synthetic(N, d) under the schedules of the batched objective:
time_fit.py N d [reps]; environment: GPRY_HIP_FIT_SCHEDULE, GPRY_HIP_FIT_TP_GROUPS, GPRY_HIP_FIT_BATCH_CONTEXTS, GPRY_TP_STREAMS"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
N, d = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bounds, X, y, Xc, truth = bench.synthetic(N, d, 16)
best = None
for rep in range(reps):
    gpr = bench.make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)
    gpr.append_to_data(X[:4], y[:4], fit_gpr=False)
    if os.environ.get("GPRY_TP_STREAMS"):
        gpr.device.set_option("lml_streams", int(os.environ["GPRY_TP_STREAMS"]))
    # share of the device calls in the wall time of the fit: the lock-step driver (scipy's routine through Python) is the rest
    from gpry_amd import _lib
    acc = {"t": 0.0, "n": 0, "calls": 0}
    orig = _lib.Device.lml_batch
    def timed(self, thetas, eval_gradient=True, _o=orig):
        t = time.perf_counter(); r = _o(self, thetas, eval_gradient); acc["t"] += time.perf_counter() - t
        acc["n"] += len(thetas); acc["calls"] += 1
        return r
    _lib.Device.lml_batch = timed
    t0 = time.perf_counter()
    gpr.append_to_data(X[4:], y[4:], fit_gpr=True)
    dt = time.perf_counter() - t0
    _lib.Device.lml_batch = orig
    if best is None or dt < best:
        best, share = dt, dict(acc)
fs = gpr.fit_stats
ev = np.array(fs["evals_per_run"])
widths = [int((ev > r).sum()) for r in range(ev.max())]        # thetas in flight per round (all groups together)
hist = np.bincount(np.minimum(widths, 48))
print("rounds by width (all groups together):", {int(w): int(c) for w, c in enumerate(hist) if c}, flush=True)
print(f"N={N} d={d} schedule={os.environ.get('GPRY_HIP_FIT_SCHEDULE', 'auto')}->{fs.get('schedule')} groups={fs.get('contexts')} "
      f"streams={os.environ.get('GPRY_TP_STREAMS', 'default')}: {best * 1e3:.0f} ms; evaluations {gpr.n_eval_loglike}, rounds {max(fs['evals_per_run'])}; "
      f"lml {gpr.log_marginal_likelihood_value_:.6f}; in gpry_lml_batch {share['t'] * 1e3:.0f} ms over {share['calls']} calls "
      f"({share['n']} thetas; summed over the groups' threads)", flush=True)
