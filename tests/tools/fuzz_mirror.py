#!/usr/bin/env python3
"""Stateful randomised comparison of the host mirror on the GPU (GaussianProcessRegressor, NORA,
RankedPool through the device) with the oracle: random sequences of appends at fixed theta,
predictions, conditioned (Kriging-believer) models, copies / pickles and multi_add calls."""
import copy
import os
import pickle
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import gpry_oracle as orc  # noqa: E402

KSPEC = {0: "RBF", 1: {"Matern": {"nu": 0.5}}, 2: {"Matern": {"nu": 1.5}}, 3: {"Matern": {"nu": 2.5}}}


def run(n_seq=6, seed=0):
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.gp_acquisition import NORA
    from gpry_amd.kernels import clone
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
    rng = np.random.default_rng(seed)
    bad = 0
    worst = {"mean": 0.0, "var": 0.0, "cond_std": 0.0, "acq": 0.0}
    for seq in range(n_seq):
        d = int(rng.integers(1, 9))
        kid = int(rng.integers(0, 4))
        bounds = np.stack([-rng.uniform(1, 4, d), rng.uniform(1, 4, d)], axis=1)
        theta = np.log(np.concatenate(([10 ** rng.uniform(0, 1.5)], 10 ** rng.uniform(-0.6, 0.0, d))))
        ref = orc.OracleGPR(bounds, kernel_id=kid, noise_level=float(10 ** rng.uniform(-2.5, -1.5)))
        ref.theta = theta.copy()
        ref.fitted = True
        gpr = GaussianProcessRegressor(kernel=KSPEC[kid], bounds=bounds, noise_level=ref.noise_level,
                                       preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
                                       account_for_inf=None)
        k = clone(gpr.kernel)
        k.theta = theta
        gpr.kernel_, gpr._fitted = k, True

        def truth(P):
            return -0.5 * ((P / (bounds[:, 1] - bounds[:, 0])) ** 2).sum(1) * 40.0

        n0 = int(rng.integers(2 * d + 2, 150))
        X0 = rng.uniform(bounds[:, 0], bounds[:, 1], (n0, d))
        ref.append_to_data(X0, truth(X0), fit_gpr=False, fit_preprocessors=True)
        gpr.append_to_data(X0, truth(X0), fit_gpr=False)
        for step in range(int(rng.integers(3, 7))):
            op = rng.choice(["append", "predict", "conditioned", "copy", "multi_add"])
            M = int(rng.choice([1, 3, 16, 17, 200, 1500]))
            Xc = rng.uniform(bounds[:, 0], bounds[:, 1], (M, d))
            C = np.exp(theta[0]) * ref.pre_y.std_ ** 2
            scale = max(1.0, np.max(np.abs(ref.y_train)))
            if op == "append":
                kx = int(rng.integers(1, d + 3))
                Xn = rng.uniform(bounds[:, 0], bounds[:, 1], (kx, d))
                ref.append_to_data(Xn, truth(Xn), fit_gpr=False, fit_preprocessors=True)
                gpr.append_to_data(Xn, truth(Xn), fit_gpr=False)
            target = gpr
            if op == "copy":
                target = copy.deepcopy(gpr) if rng.random() < 0.5 else pickle.loads(pickle.dumps(gpr))
            if op in ("predict", "copy", "append"):
                rm, rs = ref.predict(Xc, return_std=True)
                m, s = target.predict(Xc, return_std=True)
                e1 = np.max(np.abs(m - rm)) / scale
                e2 = np.max(np.abs(s ** 2 - rs ** 2)) / C
                worst["mean"], worst["var"] = max(worst["mean"], e1), max(worst["var"], e2)
                if e1 > 1e-7 or e2 > 1e-8:
                    print(f"seq {seq} {op}: mean err {e1:.2e} var err {e2:.2e} (d={d} kid={kid} n={ref.n})"); bad += 1
                e3 = np.max(np.abs(target.predict_std(Xc) ** 2 - rs ** 2)) / C
                if e3 > 1e-8:
                    print(f"seq {seq} {op}: predict_std err {e3:.2e}"); bad += 1
            elif op == "conditioned":
                kx = int(rng.integers(1, 5))
                Xl = rng.uniform(bounds[:, 0], bounds[:, 1], (kx, d))
                yl = gpr.predict(Xl)
                cg = gpr.conditioned(Xl, yl)
                cr = ref.conditioned_copy(Xl, ref.predict(Xl))
                e = np.max(np.abs(cg.predict_std(Xc) ** 2 - cr.predict_std(Xc) ** 2)) / C
                worst["cond_std"] = max(worst["cond_std"], e)
                if e > 1e-7:
                    print(f"seq {seq} conditioned: var err {e:.2e} (d={d} kid={kid} n={ref.n} k={kx})"); bad += 1
            elif op == "multi_add":
                npts = int(rng.integers(1, d + 2))
                Mc = int(rng.choice([300, 2000, 40000]))
                Xp = rng.uniform(bounds[:, 0], bounds[:, 1], (Mc, d))
                acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0)
                acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xp, None, None, None)
                Xa, ya, aa = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(1))
                Xr, yr, ar = orc.nora_multi_add(ref, Xp, npts)
                if not np.array_equal(Xa, Xr):
                    print(f"seq {seq} multi_add: proposals differ (d={d} kid={kid} n={ref.n} M={Mc} npts={npts})")
                    bad += 1
                else:
                    e = np.max(np.abs(aa - ar)) if len(aa) else 0.0
                    worst["acq"] = max(worst["acq"], e)
                    if e > 1e-5:
                        print(f"seq {seq} multi_add: acq err {e:.2e}"); bad += 1
    return bad, worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    bad, worst = run(n, seed)
    print(f"{n} sequences in {time.time() - t0:.1f} s; worst: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()) +
          f"; violations: {bad}")
    sys.exit(1 if bad else 0)
