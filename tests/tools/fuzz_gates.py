#!/usr/bin/env python3
"""Randomised check of the device gates: models with a random -inf region (SVM classifier trained by
the mirror) and a trust region; NORA proposals, y and sigma with the gates on the device must equal
those with the host-side masks (libsvm + numpy)."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

KSPEC = {0: "RBF", 2: {"Matern": {"nu": 1.5}}, 3: {"Matern": {"nu": 2.5}}}


def run(n_cases=10, seed=0):
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.gp_acquisition import NORA
    from gpry_amd.kernels import clone
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
    rng = np.random.default_rng(seed)
    bad = n_inf = 0
    for case in range(n_cases):
        d = int(rng.integers(2, 7))
        kid = int(rng.choice([0, 2, 3]))
        bounds = np.stack([-rng.uniform(2, 5, d), rng.uniform(2, 5, d)], axis=1)
        N = int(rng.integers(60, 300))
        X = rng.uniform(bounds[:, 0], bounds[:, 1], (N, d))
        y = -0.5 * (X ** 2).sum(1)
        normal = rng.standard_normal(d)
        y[X @ normal > rng.uniform(0.5, 2.0)] = -np.inf           # an "unphysical" half-space
        if np.all(np.isinf(y)) or not np.any(np.isinf(y)):
            continue
        res = []
        M = int(rng.choice([500, 5000, 60000]))
        Xc = rng.uniform(bounds[:, 0], bounds[:, 1], (M, d))
        trf = float(rng.choice([1.2, 2.0])) if case % 2 else None
        npts = int(rng.integers(1, d + 1))
        for use_device in (True, False):
            gpr = GaussianProcessRegressor(kernel=KSPEC[kid], bounds=bounds, preprocessing_X=Normalize_bounds(bounds),
                                           preprocessing_y=Normalize_y(), account_for_inf="SVM", inf_threshold="20s",
                                           trust_region_factor=trf, random_state=1)
            k = clone(gpr.kernel)
            k.theta = np.log(np.concatenate(([5.0], np.full(d, 0.4))))
            gpr.kernel_, gpr._fitted = k, True
            gpr.append_to_data(X, y, fit_gpr=False)
            if not use_device:
                gpr._push_gates = lambda *a, **k: (gpr.device.set_gates(), False)[1]
            acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0)
            acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
            out = acq.multi_add(gpr, n_points=npts, bounds=gpr.trust_bounds, rng=np.random.default_rng(0))
            ys = acq.last_MC_sample()[1]
            res.append((out[0], out[1], out[2], ys))
        n_inf += int(np.isneginf(res[0][3]).sum())
        same_mask = np.array_equal(np.isneginf(res[0][3]), np.isneginf(res[1][3]))
        if not (same_mask and all(np.array_equal(a, b) for a, b in zip(res[0][:3], res[1][:3]))):
            dmask = int((np.isneginf(res[0][3]) != np.isneginf(res[1][3])).sum())
            print(f"case {case}: device gates differ from host masks (d={d} kid={kid} N={N} M={M}; {dmask} verdicts)")
            bad += 1
    return bad, n_inf


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad, n_inf = run(n, seed)
    print(f"{n} cases in {time.time() - t0:.1f} s; {n_inf} candidates gated; violations: {bad}")
    sys.exit(1 if bad else 0)
