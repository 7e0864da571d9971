#!/usr/bin/env python3
"""Generate golden input/output vectors from the REAL reference (GPry 3.0.0).

Runs only in the build container, where ``/root/reference`` is mounted; it is a
no-op elsewhere.  It imports the reference's own Python modules (with a stub for
the missing optional ``getdist`` dependency, which is only needed by plotting/MC
code that is off the hot path), drives the reference's public API, and writes DATA
ONLY (``.npz`` arrays) under ``tests/golden/``.  No reference source is copied.

    python tools/make_goldens.py
"""
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    names = ("getdist", "getdist.mcsamples", "getdist.gaussian_mixtures")
    gd, gm, gg = (types.ModuleType(n) for n in names)
    gm.MCSamples = type("MCSamples", (), {})
    gm.loadMCSamples = lambda *a, **k: None
    gg.GaussianND = type("GaussianND", (), {})
    gd.mcsamples, gd.gaussian_mixtures = gm, gg
    sys.modules.update(dict(zip(names, (gd, gm, gg))))
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    import gpry  # noqa: F401
    return gpry.__version__


KERNELS = {0: ("RBF", {}), 1: ("Matern", {"nu": 0.5}), 2: ("Matern", {"nu": 1.5}),
           3: ("Matern", {"nu": 2.5})}


def make_gpr(bounds, kid, X, y, theta, noise_level=1e-2, clip_factor=1.1,
             trust_region_factor=None, normalize=True, account_for_inf=None):
    """Reference GPR at fixed theta (no optimiser run)."""
    from sklearn.base import clone
    from gpry.gpr import GaussianProcessRegressor
    from gpry.preprocessing import Normalize_bounds, Normalize_y
    name, kw = KERNELS[kid]
    gpr = GaussianProcessRegressor(
        kernel={name: kw}, bounds=bounds, optimizer=None, noise_level=noise_level,
        clip_factor=clip_factor, trust_region_factor=trust_region_factor,
        preprocessing_X=Normalize_bounds(bounds) if normalize else None,
        preprocessing_y=Normalize_y() if normalize else None,
        account_for_inf=account_for_inf)
    k = clone(gpr.kernel)
    k.theta = theta
    gpr.kernel_ = k
    gpr._fitted = True
    gpr.append_to_data(X, y, fit_gpr=False)
    return gpr


def gauss_problem(N, d, M, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    Lc = np.linalg.cholesky(Sigma)
    bounds = np.array([[-5.0, 5.0]] * d)
    X = np.clip(rng.standard_normal((N, d)) @ Lc.T, -5, 5)
    y = -0.5 * np.einsum("ni,ij,nj->n", X, np.linalg.inv(Sigma), X)
    Xc = np.clip(rng.standard_normal((M, d)) @ (np.sqrt(1.5) * Lc).T, -5, 5)
    return bounds, X, y, Xc


def f1_kernels(out):
    """F1: kernel values + theta gradients, incl. duplicate rows and cross-kernel."""
    from gpry.kernels import RBF, Matern, ConstantKernel as C
    rng = np.random.default_rng(101)
    for kid, (name, kw) in KERNELS.items():
        for d in (1, 2, 5):
            N, M = 14, 9
            X = rng.uniform(0, 1, (N, d))
            X[5] = X[2]  # duplicate row -> r = 0 branch (sklearn:kernels.py:1752-1761)
            Y = rng.uniform(0, 1, (M, d))
            Y[0] = X[3]
            ls = rng.uniform(0.05, 1.5, d)
            cval = float(rng.uniform(0.5, 20))
            cls = RBF if name == "RBF" else Matern
            kern = C(cval, [1e-4, 1e6]) * cls(list(ls), [1e-3, 10.], **kw)
            K, dK = kern(X, eval_gradient=True)
            Kx = kern(Y, X)
            out[f"f1_k{kid}_d{d}_X"] = X
            out[f"f1_k{kid}_d{d}_Y"] = Y
            out[f"f1_k{kid}_d{d}_theta"] = kern.theta
            out[f"f1_k{kid}_d{d}_K"] = K
            out[f"f1_k{kid}_d{d}_dK"] = dK
            out[f"f1_k{kid}_d{d}_Kx"] = Kx
            out[f"f1_k{kid}_d{d}_diag"] = kern.diag(Y)


def f2_f3_factor_lml(out):
    """F2 factor and F3 LML(+grad) at fixed theta, plus a non-PD case."""
    for kid in KERNELS:
        d, N = 3, 40
        bounds, X, y, _ = gauss_problem(N, d, 1, seed=200 + kid)
        theta = np.log(np.array([4.0, 0.3, 0.5, 0.2]))
        gpr = make_gpr(bounds, kid, X, y, theta)
        out[f"f2_k{kid}_X_"] = gpr.X_train_
        out[f"f2_k{kid}_y_"] = gpr.y_train_
        out[f"f2_k{kid}_alpha"] = gpr.alpha
        out[f"f2_k{kid}_theta"] = theta
        out[f"f2_k{kid}_L"] = gpr.L_
        out[f"f2_k{kid}_V"] = gpr.V_
        out[f"f2_k{kid}_alpha_"] = gpr.alpha_
        th2 = theta + np.array([0.3, -0.2, 0.1, 0.4])
        lml, grad = gpr.log_marginal_likelihood(th2, eval_gradient=True)
        out[f"f3_k{kid}_theta"] = th2
        out[f"f3_k{kid}_lml"] = lml
        out[f"f3_k{kid}_grad"] = grad
        out[f"f3_k{kid}_lml_nograd"] = gpr.log_marginal_likelihood(th2)
    # non-PD: duplicated points, zero noise, huge length scale (RBF)
    d, N = 2, 12
    bounds, X, y, _ = gauss_problem(N, d, 1, seed=222)
    X[1] = X[0]
    X[7] = X[3]
    theta = np.log(np.array([1.0, 10.0, 10.0]))
    gpr = make_gpr(bounds, 0, X, y, np.log(np.array([1.0, 0.3, 0.3])), noise_level=1e-2)
    gpr.alpha = np.zeros_like(gpr.alpha)
    lml, grad = gpr.log_marginal_likelihood(theta, eval_gradient=True)
    out["f3_nonpd_X_"] = gpr.X_train_
    out["f3_nonpd_y_"] = gpr.y_train_
    out["f3_nonpd_theta"] = theta
    out["f3_nonpd_lml"] = lml
    out["f3_nonpd_grad"] = grad


def f4_predict(out):
    """F4: predict mean/std with clipping, a trust region, and an SVM-style mask."""
    N, d, M = 128, 4, 512
    for kid in (0, 3):
        bounds, X, y, Xc = gauss_problem(N, d, M, seed=300 + kid)
        theta = np.log(np.array([4.0, 0.3, 0.25, 0.4, 0.35]))
        # clip_factor=1 and candidates at the mode: predictions overshoot max(y_train)
        gpr = make_gpr(bounds, kid, X, y, theta, clip_factor=1.0)
        Xc[:8] = X[np.argsort(y)[-8:]] * 0.05
        mean, std = gpr.predict(Xc, return_std=True)
        out[f"f4_k{kid}_bounds"] = bounds
        out[f"f4_k{kid}_X"] = X
        out[f"f4_k{kid}_y"] = y
        out[f"f4_k{kid}_Xc"] = Xc
        out[f"f4_k{kid}_theta"] = theta
        out[f"f4_k{kid}_mean"] = mean
        out[f"f4_k{kid}_std"] = std
        out[f"f4_k{kid}_std_only"] = gpr.predict_std(Xc)
        out[f"f4_k{kid}_clip_hi"] = max(y)
        # trust region
        gtr = make_gpr(bounds, kid, X, y, theta, trust_region_factor=0.6,
                       clip_factor=1.0)
        mean_tr, std_tr = gtr.predict(Xc, return_std=True)
        out[f"f4_k{kid}_trust_bounds"] = gtr.trust_bounds
        out[f"f4_k{kid}_mean_tr"] = mean_tr
        out[f"f4_k{kid}_std_tr"] = std_tr

    # classifier-masked case: a stand-in classifier object that marks a fixed subset
    # as infinite (the reference only calls .predict/.fit/._is_finite_raw on it).
    class FixedMaskClassifier:
        n = None
        abs_threshold = None

        def __init__(self, mask_fn):
            self.mask_fn = mask_fn

        def _is_finite_raw(self, y, diff_threshold):
            return np.full(len(y), True)

        def is_finite(self, y):
            return np.full(len(y), True)

        def fit(self, X, y, diff_threshold):
            return np.full(len(y), True)

        def predict(self, X_, validate=True):
            return self.mask_fn(X_)

    kid = 3
    bounds, X, y, Xc = gauss_problem(N, d, M, seed=311)
    theta = np.log(np.array([4.0, 0.3, 0.25, 0.4, 0.35]))
    mask_fn = lambda X_: X_[:, 0] < 0.62  # noqa: E731  (in transformed coordinates)
    gsvm = make_gpr(bounds, kid, X, y, theta,
                    account_for_inf=FixedMaskClassifier(mask_fn))
    mean, std = gsvm.predict(Xc, return_std=True)
    out["f4_svm_bounds"] = bounds
    out["f4_svm_X"] = X
    out["f4_svm_y"] = y
    out["f4_svm_Xc"] = Xc
    out["f4_svm_theta"] = theta
    out["f4_svm_finite"] = mask_fn(gsvm.preprocessing_X.transform(Xc))
    out["f4_svm_mean"] = mean
    out["f4_svm_std"] = std
    out["f4_svm_std_only"] = gsvm.predict_std(Xc)


def f5_logexp(out):
    from gpry.acquisition_functions import LogExp
    rng = np.random.default_rng(5)
    mu = rng.normal(-20, 10, 64)
    std = np.abs(rng.normal(0.5, 0.5, 64))
    std[:6] = [0.0, 1e-2, 1e-2 * (1 - 1e-12), 5e-3, 1e-2 * (1 + 1e-9), 3.0]
    mu[10:13] = -np.inf
    af = LogExp(dimension=7)
    with np.errstate(all="ignore"):
        out["f5_acq"] = LogExp.f(mu, std, baseline=-3.5, noise_level=1e-2, zeta=af.zeta)
    out["f5_mu"] = mu
    out["f5_std"] = std
    out["f5_zeta"] = af.zeta
    out["f5_baseline"] = -3.5
    out["f5_noise"] = 1e-2


def f6_fit(out):
    """F6: full fits (restarts) and a 'simple' refit; RNG order of restart starts."""
    from gpry.gpr import GaussianProcessRegressor
    from gpry.preprocessing import Normalize_bounds, Normalize_y
    for kid, N, d in ((0, 48, 2), (3, 60, 3)):
        bounds, X, y, Xc = gauss_problem(N + 8, d, 32, seed=600 + kid)
        name, kw = KERNELS[kid]
        gpr = GaussianProcessRegressor(
            kernel={name: kw}, bounds=bounds, n_restarts_optimizer=4,
            preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
            account_for_inf=None, random_state=3)
        gpr.append_to_data(X[:N], y[:N], fit_gpr=True)
        out[f"f6_k{kid}_bounds"] = bounds
        out[f"f6_k{kid}_X"] = X
        out[f"f6_k{kid}_y"] = y
        out[f"f6_k{kid}_Xc"] = Xc
        out[f"f6_k{kid}_theta_bounds"] = gpr.kernel_.bounds
        out[f"f6_k{kid}_theta_full"] = gpr.kernel_.theta
        out[f"f6_k{kid}_lml_full"] = gpr.log_marginal_likelihood_value_
        out[f"f6_k{kid}_neval_full"] = gpr.n_eval_loglike
        m, s = gpr.predict(Xc, return_std=True)
        out[f"f6_k{kid}_mean_full"] = m
        out[f"f6_k{kid}_std_full"] = s
        gpr.append_to_data(X[N:], y[N:], fit_gpr="simple")
        out[f"f6_k{kid}_theta_simple"] = gpr.kernel_.theta
        out[f"f6_k{kid}_lml_simple"] = gpr.log_marginal_likelihood_value_
        m, s = gpr.predict(Xc, return_std=True)
        out[f"f6_k{kid}_mean_simple"] = m
        out[f"f6_k{kid}_std_simple"] = s


def f7_multi_add(out):
    """F7: NORA.multi_add with an injected candidate pool (two shapes + re-use)."""
    from gpry.gp_acquisition import NORA
    for tag, kid, N, d, M, npts in (("a", 3, 96, 3, 4096, 3), ("b", 0, 160, 8, 8000, 8)):
        bounds, X, y, Xc = gauss_problem(N, d, M, seed=700 + d)
        theta = np.log(np.array([4.0] + [0.3] * d))
        gpr = make_gpr(bounds, kid, X, y, theta)
        acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None, Xc=Xc: (
            Xc, None, None, None)
        Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        Xmc, ymc, smc, _ = acq.last_MC_sample(warn_reweight=False)
        out[f"f7{tag}_bounds"] = bounds
        out[f"f7{tag}_X"] = X
        out[f"f7{tag}_y"] = y
        out[f"f7{tag}_theta"] = theta
        out[f"f7{tag}_kid"] = kid
        out[f"f7{tag}_seed"] = 700 + d
        out[f"f7{tag}_M"] = M
        out[f"f7{tag}_Xc_sum"] = Xc.sum(axis=0)
        if tag == "a":
            out["f7a_Xc"] = Xc
        out[f"f7{tag}_y_mc"] = ymc
        out[f"f7{tag}_sigma_mc"] = smc
        out[f"f7{tag}_X_pool"] = Xp
        out[f"f7{tag}_y_pool"] = yp
        out[f"f7{tag}_acq_pool"] = ap
        out[f"f7{tag}_acq_cond"] = acq.pool.acq_cond
        out[f"f7{tag}_pool_sigma"] = acq.pool.sigma
        out[f"f7{tag}_cache_counter"] = acq.pool.cache_counter
        # second call re-uses the MC sample (mc_every=2): reweighting and the
        # already-proposed filter (gp_acquisition.py:875-919, 1037-1047); the GPR has
        # meanwhile absorbed the first batch with its true values.
        Sig = None
        rng = np.random.default_rng(700 + d)
        A = rng.standard_normal((d, d))
        Sig = A @ A.T / d + 0.5 * np.eye(d)
        y_true = -0.5 * np.einsum("ni,ij,nj->n", Xp, np.linalg.inv(Sig), Xp)
        gpr.append_to_data(Xp, y_true, fit_gpr=False)
        Xp2, yp2, ap2 = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        Xr, yr, sr, wr = acq.last_MC_sample(warn_reweight=False)
        out[f"f7{tag}_y_new"] = y_true
        out[f"f7{tag}_X_pool2"] = Xp2
        out[f"f7{tag}_y_pool2"] = yp2
        out[f"f7{tag}_acq_pool2"] = ap2
        out[f"f7{tag}_n_rw"] = len(yr)
        out[f"f7{tag}_w_rw_sum"] = wr.sum()
        out[f"f7{tag}_y_rw_sum"] = yr.sum()


def f8_append(out):
    """F8: append_to_data(fit_gpr=False) -> updated L_, V_, alpha_."""
    kid, N, d = 2, 32, 3
    bounds, X, y, Xc = gauss_problem(N + 3, d, 16, seed=800)
    theta = np.log(np.array([4.0, 0.3, 0.5, 0.2]))
    gpr = make_gpr(bounds, kid, X[:N], y[:N], theta)
    s0 = gpr.predict_std(Xc)
    gpr.append_to_data(X[N:], y[N:], fit_gpr=False, fit_classifier=False)
    out["f8_bounds"] = bounds
    out["f8_X"] = X
    out["f8_y"] = y
    out["f8_Xc"] = Xc
    out["f8_theta"] = theta
    out["f8_std_before"] = s0
    out["f8_L"] = gpr.L_
    out["f8_V"] = gpr.V_
    out["f8_alpha_"] = gpr.alpha_
    out["f8_X_train_"] = gpr.X_train_
    out["f8_y_train_"] = gpr.y_train_
    m, s = gpr.predict(Xc, return_std=True)
    out["f8_mean_after"] = m
    out["f8_std_after"] = s


def f9_config1(out):
    """F9: config 1 plumbing -- curved degeneracy (tests/model_generator.py:134)."""
    sys.path.insert(0, os.path.join(REF, "tests"))
    a, b, c, dd = 10., 0.45, 4., 20.
    bounds = np.array([[-0.5, 1.5], [-0.5, 2.]])
    rng = np.random.default_rng(0)
    X = rng.uniform(bounds[:, 0], bounds[:, 1], (64, 2))
    y = -(a * (b - X[:, 0])) ** 2. / c - (dd * (X[:, 1] / c - X[:, 0] ** 4.)) ** 2.
    from gpry.gpr import GaussianProcessRegressor
    from gpry.preprocessing import Normalize_bounds, Normalize_y
    gpr = GaussianProcessRegressor(
        kernel="RBF", bounds=bounds, n_restarts_optimizer=3,
        preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
        account_for_inf=None, random_state=3)
    gpr.append_to_data(X, y, fit_gpr=True)
    Xc = rng.uniform(bounds[:, 0], bounds[:, 1], (200, 2))
    m, s = gpr.predict(Xc, return_std=True)
    out["f9_bounds"] = bounds
    out["f9_X"] = X
    out["f9_y"] = y
    out["f9_Xc"] = Xc
    out["f9_theta"] = gpr.kernel_.theta
    out["f9_lml"] = gpr.log_marginal_likelihood_value_
    out["f9_mean"] = m
    out["f9_std"] = s


def f10_gradients(out):
    """F10: predict(..., return_mean_grad, return_std_grad) and the LogExp gradient for single
    points (one of them a training point: the r = 0 branches), all four kernels."""
    from gpry.acquisition_functions import LogExp
    N, d, M = 72, 3, 6
    bounds, X, y, Xc = gauss_problem(N, d, M, seed=21)
    Xc[1] = X[5]                       # exactly on a training point
    out["f10_bounds"], out["f10_X"], out["f10_y"], out["f10_Xc"] = bounds, X, y, Xc
    for kid in KERNELS:
        theta = np.log(np.array([3.0, 0.35, 0.5, 0.25]))
        gpr = make_gpr(bounds, kid, X, y, theta)
        af = LogExp(dimension=d)
        res = {k: [] for k in ("mean", "std", "mean_grad", "std_grad", "acq", "acq_grad", "kgrad")}
        if kid == 1:
            # Matern-1/2: Product.gradient_x hands a (1, d) point to Matern.gradient_x, whose
            # r = 0 fill is shaped for a (d,) point (gpry/kernels.py:355-359) and raises.  Pin
            # the factor kernel called the way its docstring says (1-d point) instead.
            try:
                gpr.predict(Xc[:1], return_std=True, return_mean_grad=True, return_std_grad=True)
                out["f10_k1_reference_raises"] = np.array(0)
            except ValueError:
                out["f10_k1_reference_raises"] = np.array(1)
            kg = []
            for x in Xc:
                x_ = gpr.preprocessing_X.transform(x[None, :])[0]
                kg.append(gpr.kernel_.k1.constant_value * gpr.kernel_.k2.gradient_x(x_, gpr.X_train_))
            out["f10_k1_theta"] = theta
            out["f10_k1_kgrad"] = np.array(kg)
            continue
        for x in Xc:
            m, s, mg, sg = gpr.predict(x[None, :], return_std=True, return_mean_grad=True,
                                       return_std_grad=True)
            a, ag = af(x[None, :], gpr, eval_gradient=True)
            x_ = gpr.preprocessing_X.transform(x[None, :])[0]
            res["kgrad"].append(gpr.kernel_.gradient_x(x_, gpr.X_train_))
            for k, v in zip(("mean", "std", "mean_grad", "std_grad", "acq", "acq_grad"),
                            (m[0], s[0], mg, sg, a[0], ag)):
                res[k].append(np.array(v, dtype=float))
        p = f"f10_k{kid}_"
        out[p + "theta"] = theta
        out[p + "zeta"] = af.zeta
        for k, v in res.items():
            out[p + k] = np.array(v)


def f10b_batch_optimizer(out):
    """F10b: BatchOptimizer (gpry/gp_acquisition.py:127-525) -- optimize_acquisition_function runs from the
    last training point and from proposed starts, the lie-append loop of multi_add -- with a uniform
    proposer and a seeded generator; plus predict-with-gradients for a batch of points, one reference call
    per point (the reference has no batched form)."""
    from gpry.gp_acquisition import BatchOptimizer
    from gpry.proposal import UniformProposer
    N, d = 60, 3
    bounds, X, y, Xq = gauss_problem(N, d, 9, seed=33)
    Xq[2] = X[7]
    theta = np.log(np.array([3.0, 0.35, 0.5, 0.25]))
    out["f10b_bounds"], out["f10b_X"], out["f10b_y"], out["f10b_Xq"], out["f10b_theta"] = bounds, X, y, Xq, theta
    gpr = make_gpr(bounds, 3, X, y, theta)
    res = {k: [] for k in ("mean", "std", "mean_grad", "std_grad")}
    for x in Xq:
        vals = gpr.predict(x[None, :], return_std=True, return_mean_grad=True, return_std_grad=True)
        for k, v in zip(("mean", "std", "mean_grad", "std_grad"), (vals[0][0], vals[1][0], vals[2], vals[3])):
            res[k].append(np.array(v, dtype=float))
    for k, v in res.items():
        out["f10b_" + k] = np.array(v)
    gpr = make_gpr(bounds, 3, X, y, theta)
    acq = BatchOptimizer(bounds, proposer=UniformProposer(bounds), n_restarts_optimizer=3,
                         n_repeats_propose=2, verbose=0)
    Xo, yl, av = acq.multi_add(gpr, n_points=3, rng=np.random.default_rng(5))
    out["f10b_X_opts"], out["f10b_y_lies"], out["f10b_acq_vals"] = Xo, yl, av
    out["f10b_n_eval"] = np.array(gpr.n_eval)
    # one optimiser run on its own: from the last training point (i = 0) and from proposals (i = 1)
    gpr = make_gpr(bounds, 3, X, y, theta)
    acq = BatchOptimizer(bounds, proposer=UniformProposer(bounds), n_restarts_optimizer=2,
                         n_repeats_propose=1, verbose=0)
    rng = np.random.default_rng(9)
    x0, f0 = acq.optimize_acquisition_function(gpr, 0, bounds=bounds, rng=rng)
    x1, f1 = acq.optimize_acquisition_function(gpr, 1, bounds=bounds, rng=rng)
    out["f10b_opt_x"], out["f10b_opt_f"] = np.array([x0, x1]), np.array([float(f0), float(f1)])


def f6b_fit_mid(out):
    """F6b: multi-restart fits of a few hundred points -- the sizes at which the device steps the restarts side by side on
    the batched kernel chain (gpry_lml_batch above N = 128)."""
    from gpry.gpr import GaussianProcessRegressor
    from gpry.preprocessing import Normalize_bounds, Normalize_y
    for kid, N, d in ((3, 200, 3), (0, 300, 4)):
        bounds, X, y, Xc = gauss_problem(N, d, 32, seed=650 + kid)
        name, kw = KERNELS[kid]
        gpr = GaussianProcessRegressor(
            kernel={name: kw}, bounds=bounds, n_restarts_optimizer=4,
            preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
            account_for_inf=None, random_state=3)
        gpr.append_to_data(X, y, fit_gpr=True)
        p = f"f6b_k{kid}_"
        out[p + "bounds"], out[p + "X"], out[p + "y"], out[p + "Xc"] = bounds, X, y, Xc
        out[p + "theta_full"] = gpr.kernel_.theta
        out[p + "lml_full"] = gpr.log_marginal_likelihood_value_
        out[p + "neval_full"] = gpr.n_eval_loglike
        m, s = gpr.predict(Xc, return_std=True)
        out[p + "mean_full"], out[p + "std_full"] = m, s


def main():
    if not os.path.isdir(REF):
        print("reference not mounted; nothing to do")
        return 0
    version = import_reference()
    import sklearn
    import scipy
    os.makedirs(OUT, exist_ok=True)
    groups = {"kernels": [f1_kernels], "factor_lml": [f2_f3_factor_lml],
              "predict": [f4_predict, f5_logexp, f8_append], "fit": [f6_fit, f9_config1],
              "multi_add": [f7_multi_add], "gradients": [f10_gradients, f10b_batch_optimizer],
              "fit_mid": [f6b_fit_mid]}
    for name, fns in groups.items():
        out = {}
        for fn in fns:
            fn(out)
        out["_versions"] = np.array([f"gpry {version}", f"sklearn {sklearn.__version__}",
                                     f"scipy {scipy.__version__}",
                                     f"numpy {np.__version__}"])
        path = os.path.join(OUT, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
