"""CPU restatement (numpy/scipy) of GPry's GP-regression + NORA acquisition path.

THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle and the ``cpu_baseline``
("port") of ``bench.py``; the product (``gpry_amd``) never imports it.

Parity status: PINNED.  ``tools/make_goldens.py`` imports the real reference
(``/root/reference``, GPry 3.0.0 on scikit-learn 1.7.2 / scipy 1.15.3) in the build
container and stores input/output vectors under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function below against them.

The arithmetic the reference inherits from third-party code is restated from
scikit-learn 1.7.2 (unpinned dependency of the reference, ``pyproject.toml:30-33``):
``sklearn/gaussian_process/kernels.py`` and ``_gpr.py``.  Citations below are
``file:line`` under ``/root/reference`` unless prefixed ``sklearn:``.

The routines call the same BLAS/LAPACK entry points, in the same order, as the
reference does (pdist/cdist -> elementwise -> cholesky -> solve_triangular ->
cho_solve -> dtrmm -> einsum), so timing this file on the host is a fair stand-in
for timing the reference (which cannot travel to the GPU box).
"""

import copy
import math
import warnings
from numbers import Number

import numpy as np
import scipy.optimize
from scipy.linalg import cholesky, cho_solve, solve_triangular
from scipy.linalg.blas import dtrmm
from scipy.spatial.distance import cdist, pdist, squareform

# kernel ids shared with include/gpry_hip.h
RBF, MATERN12, MATERN32, MATERN52 = 0, 1, 2, 3
KERNEL_NAMES = {RBF: "rbf", MATERN12: "matern12", MATERN32: "matern32",
                MATERN52: "matern52"}
_NU = {MATERN12: 0.5, MATERN32: 1.5, MATERN52: 2.5}


# ----------------------------------------------------------------------------
# a1 / a5 / a8: kernel values and theta-gradients
# ----------------------------------------------------------------------------
def _corr_from_dist(dist, kernel_id):
    """Correlation k(r) from the *scaled* distance array.

    ``dist`` is squared-euclidean for RBF (sklearn:kernels.py:1553-1565) and
    euclidean for Matern (sklearn:kernels.py:1708-1723).
    """
    if kernel_id == RBF:
        return np.exp(-0.5 * dist)
    if kernel_id == MATERN12:
        return np.exp(-dist)
    if kernel_id == MATERN32:
        t = dist * math.sqrt(3)
        return (1.0 + t) * np.exp(-t)
    if kernel_id == MATERN52:
        t = dist * math.sqrt(5)
        return (1.0 + t + t ** 2 / 3.0) * np.exp(-t)
    raise ValueError(f"unknown kernel id {kernel_id}")


def kernel_matrix(X, theta, kernel_id, Y=None, eval_gradient=False):
    """``C * k(X, Y)`` for theta = [log C, log l_1..l_d].

    Follows ``Product.__call__`` (sklearn:kernels.py:931-966) of ``ConstantKernel``
    (sklearn:kernels.py:1239-1291) and ``RBF``/``Matern``; theta order from
    sklearn:kernels.py:734-747.  With ``Y is None`` the diagonal of k is forced to 1
    (sklearn:kernels.py:1560,1738).  ``eval_gradient`` returns d K / d theta, shape
    (N, N, 1+d): Constant sklearn:kernels.py:1278-1289, RBF :1574-1580,
    Matern :1740-1768, product rule :960-964.
    """
    X = np.atleast_2d(X)
    theta = np.asarray(theta, dtype=float)
    const = math.exp(theta[0])
    ls = np.exp(theta[1:])
    metric = "sqeuclidean" if kernel_id == RBF else "euclidean"
    if Y is None:
        dist = pdist(X / ls, metric=metric)
        k = squareform(_corr_from_dist(dist, kernel_id))
        np.fill_diagonal(k, 1)
    else:
        if eval_gradient:
            raise ValueError("Gradient can only be evaluated when Y is None.")
        dist = cdist(X / ls, np.atleast_2d(Y) / ls, metric=metric)
        k = _corr_from_dist(dist, kernel_id)
    K = np.full(k.shape, const) * k
    if not eval_gradient:
        return K
    # per-dimension squared scaled differences (N, N, d)
    D = (X[:, None, :] - X[None, :, :]) ** 2 / (ls ** 2)
    if kernel_id == RBF:
        k_grad = D * k[..., None]
    elif kernel_id == MATERN12:
        denom = np.sqrt(D.sum(axis=2))[:, :, None]
        q = np.zeros_like(D)
        np.divide(D, denom, out=q, where=denom != 0)
        k_grad = k[..., None] * q
    elif kernel_id == MATERN32:
        k_grad = 3 * D * np.exp(-np.sqrt(3 * D.sum(-1)))[..., None]
    else:
        t = np.sqrt(5 * D.sum(-1))[..., None]
        k_grad = 5.0 / 3.0 * D * (t + 1) * np.exp(-t)
    dK = np.dstack((np.full(k.shape, const)[:, :, None] * k[:, :, None],
                    k_grad * np.full(k.shape, const)[:, :, None]))
    return K, dK


def kernel_diag(X, theta):
    """``kernel_.diag(X)`` = C (sklearn:kernels.py:985,1310,485)."""
    return np.full(np.atleast_2d(X).shape[0], math.exp(theta[0]))


# ----------------------------------------------------------------------------
# a3: factor ; a4: log marginal likelihood
# ----------------------------------------------------------------------------
def factorize(K, y_):
    """``_kernel_inverse`` (gpry/gpr.py:1453-1465): L, V = L^-1, alpha_ = K^-1 y."""
    L = cholesky(K, lower=True)
    V = solve_triangular(L, np.eye(L.shape[0]), lower=True)
    alpha_ = cho_solve((L, True), y_)
    return L, V, alpha_


def log_marginal_likelihood(X_, y_, alpha, theta, kernel_id, eval_gradient=False):
    """sklearn:_gpr.py:574-652 as reached from gpry/gpr.py:876-881.

    Non-PD returns ``(-inf, zeros)`` (sklearn:_gpr.py:586-589).
    """
    theta = np.asarray(theta, dtype=float)
    if eval_gradient:
        K, dK = kernel_matrix(X_, theta, kernel_id, eval_gradient=True)
    else:
        K = kernel_matrix(X_, theta, kernel_id)
    K[np.diag_indices_from(K)] += alpha
    try:
        L = cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return (-np.inf, np.zeros_like(theta)) if eval_gradient else -np.inf
    y2 = y_[:, None]
    a = cho_solve((L, True), y2, check_finite=False)
    lml = -0.5 * np.einsum("ik,ik->k", y2, a)
    lml -= np.log(np.diag(L)).sum()
    lml -= K.shape[0] / 2 * np.log(2 * np.pi)
    lml = lml.sum(axis=-1)
    if not eval_gradient:
        return lml
    inner = np.einsum("ik,jk->ijk", a, a)
    K_inv = cho_solve((L, True), np.eye(K.shape[0]), check_finite=False)
    inner -= K_inv[..., None]
    grad = 0.5 * np.einsum("ijl,jik->kl", inner, dK)
    return lml, grad.sum(axis=-1)


def log_marginal_likelihood_blocked(X_, y_, alpha, theta, kernel_id, block=256):
    """Same value and gradient as ``log_marginal_likelihood(..., eval_gradient=True)`` without the
    ``(N, N, 1+d)`` tensors of sklearn:_gpr.py:625-649 / sklearn:kernels.py:960-964,1574-1580,1740-1768
    (10 GB at N=4096, d=16; 56 GB at N=8192, d=20): the trace
    ``grad_k = 1/2 sum_ij (a a^T - K^-1)_ij dK_ij/dtheta_k`` (:643-645) is accumulated over blocks of
    ``block`` rows, each block's ``dK`` rows built with the formulas of ``kernel_matrix``.  Same BLAS/LAPACK
    factorisation calls; only the summation order of the trace differs (row blocks).  Used by the
    full-size parity tests of the device objective (N=8192) and checked against the einsum form on the
    F3 goldens (tests/test_oracle_golden.py)."""
    theta = np.asarray(theta, dtype=float)
    X = np.atleast_2d(X_)
    N = X.shape[0]
    const = math.exp(theta[0])
    ls = np.exp(theta[1:])
    K = kernel_matrix(X, theta, kernel_id)
    k = K / const                       # correlation with unit diagonal (sklearn:kernels.py:1560,1738)
    np.fill_diagonal(k, 1)
    K[np.diag_indices_from(K)] += alpha
    try:
        L = cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return -np.inf, np.zeros_like(theta)
    a = cho_solve((L, True), y_, check_finite=False)
    lml = -0.5 * float(y_ @ a) - np.log(np.diag(L)).sum() - N / 2 * np.log(2 * np.pi)
    W = cho_solve((L, True), np.eye(N), check_finite=False)       # K^-1
    W *= -1.0
    W += np.outer(a, a)                                           # inner = a a^T - K^-1 (:641-642)
    del K, L
    grad = np.zeros(len(theta))
    Xs = X / ls
    for i0 in range(0, N, block):
        i1 = min(N, i0 + block)
        Wb, kb = W[i0:i1], k[i0:i1]
        grad[0] += 0.5 * np.sum(Wb * (const * kb))                # d K / d log C = K (:1278-1289)
        D = (Xs[i0:i1, None, :] - Xs[None, :, :]) ** 2            # (b, N, d) = (x_i - x_j)^2 / l^2
        if kernel_id == RBF:
            g = D * kb[..., None]
        elif kernel_id == MATERN12:
            denom = np.sqrt(D.sum(axis=2))[:, :, None]
            q = np.zeros_like(D)
            np.divide(D, denom, out=q, where=denom != 0)
            g = kb[..., None] * q
        elif kernel_id == MATERN32:
            g = 3 * D * np.exp(-np.sqrt(3 * D.sum(-1)))[..., None]
        else:
            t = np.sqrt(5 * D.sum(-1))[..., None]
            g = 5.0 / 3.0 * D * (t + 1) * np.exp(-t)
        grad[1:] += 0.5 * const * np.einsum("ij,ijk->k", Wb, g)
    return lml, grad


# ----------------------------------------------------------------------------
# f3: x-gradients
# ----------------------------------------------------------------------------
def kernel_gradient_x(x_, X_train_, theta, kernel_id):
    """``kernel_.gradient_x(x, X_train)`` of ``C * RBF|Matern`` (gpry/kernels.py:257-278,
    326-432, product rule :687-699): d k(x, X_j) / d x, shape (N, d), in the coordinates the
    kernel sees (the reference never applies the chain rule of ``Normalize_bounds``)."""
    x_ = np.asarray(x_, dtype=float)
    X_train_ = np.asarray(X_train_, dtype=float)
    const = math.exp(theta[0])
    ls = np.exp(np.asarray(theta[1:], dtype=float))
    diff = (x_ - X_train_) / ls
    dist_sq = np.sum(diff ** 2, axis=1)
    dist = np.sqrt(dist_sq)
    nz = dist != 0.0
    if kernel_id == RBF:
        g = -np.exp(-0.5 * dist_sq)[:, None] * diff / ls
    elif kernel_id == MATERN12:
        g = -np.ones_like(diff)                      # x == X_j: the reference's -1 / l
        e = -np.exp(-dist)
        g[nz] = (e[nz] / dist[nz])[:, None] * diff[nz]
        g = g / ls
    elif kernel_id == MATERN32:
        by = np.zeros_like(dist)
        by[nz] = math.sqrt(3) / dist[nz]
        f_grad = diff / ls * by[:, None]
        gexp = np.exp(-math.sqrt(3) * dist)[:, None]
        g = gexp * f_grad * (-(math.sqrt(3) * dist))[:, None]
    else:
        f = (1 + math.sqrt(5) * dist + (5.0 / 3.0) * dist_sq)[:, None]
        rec = np.zeros_like(dist)
        rec[nz] = math.sqrt(5) / dist[nz]
        d2 = diff / ls
        f1_grad = rec[:, None] * d2
        f_grad = f1_grad + (10.0 / 3.0) * d2
        gexp = np.exp(-math.sqrt(5) * dist)[:, None]
        g = f * (-gexp * f1_grad) + gexp * f_grad
    return const * g


def logexp_gradient(std, mu_grad, std_grad, sigma_n, zeta):
    """Gradient branch of ``BaseLogExp.__call__`` for one point
    (gpry/acquisition_functions.py:993-1007): ``std_grad / (std - sigma_n) + 2 zeta mu_grad``,
    ``+inf`` where ``std <= sigma_n``."""
    if std > sigma_n:
        return std_grad / (std - sigma_n) + 2 * zeta * mu_grad
    return np.ones_like(std_grad) * np.inf


# ----------------------------------------------------------------------------
# a12: LogExp
# ----------------------------------------------------------------------------
def logexp_f(mu, std, baseline, noise_level, zeta):
    """``LogExp.f`` (gpry/acquisition_functions.py:1068-1074)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return (2 * zeta * (mu - baseline) +
                np.log(np.sqrt(np.clip(std ** 2. - noise_level ** 2., 0., None))))


def auto_zeta(dimension, scaling=0.85):
    """gpry/acquisition_functions.py:933-934."""
    return dimension ** (-scaling)


# ----------------------------------------------------------------------------
# a11: preprocessors (gpry/preprocessing.py:311-411, 528-630)
# ----------------------------------------------------------------------------
class NormalizeBounds:
    def __init__(self, bounds):
        bounds = np.asarray(bounds, dtype=float)
        self.lo, self.hi = bounds[:, 0], bounds[:, 1]

    def transform(self, X):
        return (X - self.lo) / (self.hi - self.lo)

    def transform_bounds(self, bounds):
        out = np.ones_like(np.asarray(bounds, dtype=float))
        out[:, 0] = 0
        return out


class NormalizeY:
    def __init__(self):
        self.mean_, self.std_ = None, None

    def fit(self, y):
        y = y[np.isfinite(y)]
        self.mean_, self.std_ = np.mean(y), np.std(y)

    def transform(self, y):
        return (y - self.mean_) / self.std_

    def inverse_transform(self, y):
        return (y * self.std_) + self.mean_

    def transform_scale(self, s):
        return s / self.std_

    def inverse_transform_scale(self, s):
        return s * self.std_


class _Identity:
    mean_, std_ = 0.0, 1.0

    def fit(self, *a):
        pass

    def transform(self, v):
        return v

    inverse_transform = transform_scale = inverse_transform_scale = transform

    def transform_bounds(self, b):
        return b


def is_in_bounds(points, bounds):
    """gpry/tools.py:263-290."""
    points = np.atleast_2d(points)
    return np.all((points >= bounds[:, 0]) & (points <= bounds[:, 1]), axis=1)


def shrink_bounds(bounds, samples, factor=1):
    """gpry/tools.py:308-360."""
    out = np.empty(shape=bounds.shape, dtype=float)
    out[:, 0] = samples.min(axis=0)
    out[:, 1] = samples.max(axis=0)
    width = out[:, 1] - out[:, 0]
    delta = (factor - 1) / 2 * width
    out[:, 0] -= delta
    out[:, 1] += delta
    out[:, 0] = np.array([out[:, 0], bounds[:, 0]]).max(axis=0)
    out[:, 1] = np.array([out[:, 1], bounds[:, 1]]).min(axis=0)
    return out


# ----------------------------------------------------------------------------
# GaussianProcessRegressor restatement (account_for_inf=None path)
# ----------------------------------------------------------------------------
class OracleGPR:
    """The slice of ``gpry.gpr.GaussianProcessRegressor`` that sits on the hot path.

    Covers: auto kernel construction (gpry/gpr.py:343-363), ``append_to_data``
    (:577-753, ``account_for_inf=None`` branch), ``fit_gpr_hyperparameters``
    (:883-994), ``_update_model`` (:996-1020), ``predict`` (:1022-1273, value and std
    branches), ``predict_std`` (:1275-1352), trust region (:554-575, nstd=None).
    """

    def __init__(self, bounds, kernel_id=RBF, output_scale_prior=(1e-2, 1e3),
                 length_scale_prior=(1e-3, 1e1), noise_level=1e-2, clip_factor=1.1,
                 n_restarts_optimizer=0, normalize_X=True, normalize_y=True,
                 trust_region_factor=None, random_state=None, optimizer="fmin_l_bfgs_b"):
        self.bounds = np.asarray(bounds, dtype=float)
        self.d = self.bounds.shape[0]
        self.kernel_id = kernel_id
        self.noise_level = noise_level
        self.clip_factor = clip_factor
        self.n_restarts_optimizer = n_restarts_optimizer
        self.optimizer = optimizer
        self.random_state = random_state
        self.trust_region_factor = trust_region_factor
        self.trust_bounds = None
        self.pre_X = NormalizeBounds(self.bounds) if normalize_X else _Identity()
        self.pre_y = NormalizeY() if normalize_y else _Identity()
        c0 = math.sqrt(output_scale_prior[0] * output_scale_prior[1]) ** 2
        l0 = math.sqrt(length_scale_prior[0] * length_scale_prior[1])
        self.theta0 = np.log(np.array([c0] + [l0] * self.d))
        self.theta_bounds = np.log(np.array(
            [[output_scale_prior[0] ** 2, output_scale_prior[1] ** 2]] +
            [list(length_scale_prior)] * self.d))
        self.theta = None  # kernel_.theta
        self.X_train = np.empty((0, self.d))
        self.y_train = np.empty((0,))
        self.X_train_ = self.y_train_ = None
        self.alpha = None
        self.L_ = self.V_ = self.alpha_ = None
        self.fitted = False
        self.n_eval = self.n_eval_loglike = 0
        self.newly_appended = 0
        self.log_marginal_likelihood_value_ = None
        self.minus_inf_value = -np.inf

    # -- properties the callers use (gpry/gpr.py:391-414)
    @property
    def n(self):
        return len(self.y_train)

    @property
    def y_max(self):
        return np.max(self.y_train)

    def lml(self, theta, eval_gradient=False):
        self.n_eval_loglike += 1
        self.theta = np.array(theta, dtype=float)  # clone_kernel=False mutates kernel_
        return log_marginal_likelihood(self.X_train_, self.y_train_, self.alpha,
                                       theta, self.kernel_id, eval_gradient)

    def append_to_data(self, X, y, fit_gpr=True, fit_preprocessors=None):
        """gpry/gpr.py:577-753 with ``infinities_classifier is None``."""
        kwargs = None
        if fit_gpr is True:
            kwargs = {}
        elif str(fit_gpr) == "simple":
            kwargs = {"simple": True}
        elif isinstance(fit_gpr, dict):
            kwargs = dict(fit_gpr)
        elif fit_gpr is not False:
            raise ValueError("bad fit_gpr")
        if fit_preprocessors is None:
            fit_preprocessors = kwargs is not None
        force = False
        if X is None and y is None:
            X, y = np.empty((0, self.d)), np.empty((0,))
            force = kwargs is not None
        n_new = len(y)
        self.X_train = np.append(self.X_train, np.atleast_2d(X).reshape(-1, self.d),
                                 axis=0)
        self.y_train = np.append(self.y_train, y)
        if fit_preprocessors:
            self.pre_y.fit(self.y_train)
        if not n_new and not force:
            return self
        self.X_train_ = self.pre_X.transform(self.X_train)
        self.y_train_ = self.pre_y.transform(self.y_train)
        nl = (np.full(len(self.y_train_), self.noise_level)
              if isinstance(self.noise_level, Number) else self.noise_level)
        self.alpha = self.pre_y.transform_scale(nl) ** 2
        self.newly_appended = n_new
        if kwargs is not None:
            self.fit_gpr_hyperparameters(**kwargs)
        else:
            self._update_model()
        if self.trust_region_factor is not None:
            self.trust_bounds = shrink_bounds(self.bounds, self.X_train,
                                              factor=self.trust_region_factor)
        return self

    def restart_starts(self, n_restarts, start_from_current):
        """Start points in the reference's RNG order (gpry/gpr.py:969-978)."""
        from sklearn.utils import check_random_state
        rs = self.random_state
        rng = rs if isinstance(rs, np.random.Generator) else check_random_state(rs)
        starts = []
        for it in range(n_restarts):
            if it == 0 and start_from_current:
                starts.append(np.array(self.theta))
            else:
                starts.append(rng.uniform(self.theta_bounds[:, 0],
                                          self.theta_bounds[:, 1]))
        return starts

    def fit_gpr_hyperparameters(self, simple=False, start_from_current=True,
                                n_restarts=None):
        if simple:
            start_from_current, n_restarts = True, 1
        if not self.fitted:
            start_from_current = False
        if n_restarts is None:
            n_restarts = self.n_restarts_optimizer
        if self.theta is None:
            self.theta = self.theta0.copy()
        if self.optimizer is None or n_restarts <= 0:
            self.log_marginal_likelihood_value_ = self.lml(self.theta)
            self._update_model()
            return self

        def obj(theta):
            val, grad = self.lml(theta, eval_gradient=True)
            return -val, -grad

        optima = []
        for th0 in self.restart_starts(n_restarts, start_from_current):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                res = scipy.optimize.minimize(obj, th0, method="L-BFGS-B", jac=True,
                                              bounds=self.theta_bounds)
            optima.append((res.x, res.fun))
        vals = [o[1] for o in optima]
        self.log_marginal_likelihood_value_ = -np.min(vals)
        self.theta = np.array(optima[int(np.argmin(vals))][0])
        self._update_model()
        self.fitted = True
        return self

    def _update_model(self):
        if self.newly_appended < 1:
            warnings.warn("No new points have been appended to the model.")
            return self
        K = kernel_matrix(self.X_train_, self.theta, self.kernel_id)
        K[np.diag_indices_from(K)] += self.alpha
        self.L_, self.V_, self.alpha_ = factorize(K, self.y_train_)
        self.newly_appended = 0
        return self

    def clip_hi(self):
        return (self.clip_factor * max(self.y_train) -
                (self.clip_factor - 1) * min(self.y_train))

    def predict(self, X, return_std=False, ignore_trust_region=False):
        """gpry/gpr.py:1022-1273 (no classifier, no gradients)."""
        X = np.atleast_2d(X)
        self.n_eval += len(X)
        outside = None
        if self.trust_bounds is not None and not ignore_trust_region:
            outside = np.logical_not(is_in_bounds(X, self.trust_bounds))
        X_ = self.pre_X.transform(X)
        K_trans = kernel_matrix(X_, self.theta, self.kernel_id, Y=self.X_train_)
        y_mean = self.pre_y.inverse_transform(K_trans.dot(self.alpha_))
        if self.clip_factor is not None:
            y_mean = np.clip(y_mean, None, self.clip_hi())
        if outside is not None:
            y_mean[outside] = self.minus_inf_value
        if not return_std:
            return y_mean
        return y_mean, self._std_from_ktrans(K_trans, X_)

    def _std_from_ktrans(self, K_trans, X_):
        M = dtrmm(1., self.V_, K_trans.T, lower=True)
        var = kernel_diag(X_, self.theta)
        var -= np.einsum("ji,ji->i", M, M, optimize=True)
        var[var < 0] = 0.0
        return self.pre_y.inverse_transform_scale(np.sqrt(var))

    def predict_with_grad(self, x):
        """``predict(x[None], return_std=True, return_mean_grad=True, return_std_grad=True)``
        for one point (gpry/gpr.py:1236-1266): gradients with respect to the transformed
        coordinates, the mean's scaled once and the std's twice by ``std_y``."""
        X = np.atleast_2d(x)
        y_mean, y_std = self.predict(X, return_std=True)
        X_ = self.pre_X.transform(X)
        K_trans = kernel_matrix(X_, self.theta, self.kernel_id, Y=self.X_train_)
        grad = kernel_gradient_x(X_[0], self.X_train_, self.theta, self.kernel_id)
        grad_mean = self.pre_y.inverse_transform_scale(grad.T.dot(self.alpha_))
        grad_std = np.zeros(X_.shape[1])
        if not np.allclose(y_std, grad_std):
            y_std_untransformed = self.pre_y.transform_scale(y_std)
            grad_std = -np.dot(K_trans, np.dot(self.V_.T.dot(self.V_), grad))[0] / y_std_untransformed
            grad_std = self.pre_y.inverse_transform_scale(self.pre_y.inverse_transform_scale(grad_std))
        return y_mean, y_std, grad_mean, grad_std

    def predict_std(self, X):
        """gpry/gpr.py:1275-1352."""
        X = np.atleast_2d(X)
        self.n_eval += len(X)
        X_ = self.pre_X.transform(X)
        K_trans = kernel_matrix(X_, self.theta, self.kernel_id, Y=self.X_train_)
        return self._std_from_ktrans(K_trans, X_)

    def conditioned_copy(self, X, y):
        """``deepcopy(gpr)`` + ``append_to_data(fit_gpr=False, fit_classifier=False)``
        as done by ``RankedPool.cache_model`` (gpry/gp_acquisition.py:1550-1553)."""
        c = copy.deepcopy(self)
        c.append_to_data(X, y, fit_gpr=False, fit_preprocessors=False)
        return c


# ----------------------------------------------------------------------------
# a14: RankedPool (Kriging-believer ranking), gpry/gp_acquisition.py:1194-1670
# ----------------------------------------------------------------------------
class OracleRankedPool:
    """Restatement of ``RankedPool`` for the "single sort acq" and "bulk" methods."""

    def __init__(self, size, gpr, acq_func):
        self.gpr, self.f = gpr, acq_func
        self.X = np.zeros((size + 1, gpr.d))
        self.y = np.zeros(size + 1)
        self.sigma = np.zeros(size + 1)
        self.acq = np.zeros(size + 1)
        self.acq_cond = np.full(size + 1, -np.inf)  # -inf marks an empty slot (:1225)
        self.models = [None] * (size + 1)
        self.cache_counter = 0
        self.n_examined = 0

    def __len__(self):
        return len(self.y) - 1

    @property
    def min_acq(self):
        return self.acq_cond[len(self) - 1]

    def cache_model(self, i):
        """:1522-1555."""
        if i < 0:
            return self.gpr
        self.models[i] = self.gpr.conditioned_copy(self.X[:i + 1], self.y[:i + 1])
        self.cache_counter += 1
        return self.models[i]

    def add(self, X, y, sigma, acq, method="single sort acq"):
        """:1290-1335."""
        X = np.atleast_2d(X)
        if method == "bulk":
            return self.add_bulk(X, y, sigma, acq)
        order = range(len(X))
        if "sort" in method:
            order = np.argsort({"acq": acq, "y": y}[method.split()[-1]])[::-1]
        for i in order:
            self.add_one(X[i], y[i], sigma[i], acq[i])

    def add_bulk(self, X, y, sigma, acq, i_start=0):
        """:1337-1390."""
        while True:
            if i_start == 0:
                acq_cond = np.asarray(acq)
            else:
                g = self.cache_model(i_start - 1)
                acq_cond = self.f(y, g.predict_std(X))
            if acq_cond.size == 0:
                return
            i_max = int(np.argmax(acq_cond))
            if acq_cond[i_max] == np.inf:
                return
            self.X[i_start], self.y[i_start] = X[i_max], y[i_max]
            self.sigma[i_start], self.acq[i_start] = sigma[i_max], acq[i_max]
            self.acq_cond[i_start] = acq_cond[i_max]
            if i_start == len(self) - 1:
                return
            keep = np.logical_not(acq_cond == -np.inf)
            keep[i_max] = False
            X, y, sigma, acq = X[keep], y[keep], sigma[keep], acq[keep]
            i_start += 1

    def add_one(self, X, y, sigma, acq):
        """:1392-1520."""
        if acq <= self.min_acq:  # early-out (:1432)
            return
        self.n_examined += 1
        if np.isnan(acq):
            raise ValueError(f"Acquisition function value not a number: {acq}")
        X2 = np.atleast_2d(X)
        n = len(self)
        i_prev = n
        a_cond = acq
        while True:
            # climb from the bottom, skipping the buffer slot (:1464-1474)
            i_new = 0
            for i in range(n):
                if self.acq_cond[-(i + 2)] >= a_cond:
                    i_new = n - i
                    break
            if i_new in (0, i_prev, n):
                break
            s_cond = self.models[i_new - 1].predict_std(X2)[0]
            a_cond = min(a_cond, self.f(y, s_cond))
            i_prev = i_new
        if i_new >= n:
            return
        for arr, val in ((self.X, X), (self.y, y), (self.sigma, sigma),
                         (self.acq, acq), (self.acq_cond, a_cond)):
            arr[i_new + 1:] = arr[i_new:-1]
            arr[i_new] = val
        self.sort(i_new + 1)
        self.acq_cond[-1] = -np.inf

    def sort(self, i_start=0):
        """:1598-1670 (tail recursion written as a loop)."""
        while i_start < len(self):
            upper = self.cache_model(i_start - 1)
            if self.acq_cond[i_start] == -np.inf:
                return
            finite = np.flatnonzero(self.acq_cond == -np.inf)
            i_end = finite[0] if len(finite) else len(self) + 1
            s_cond = upper.predict_std(self.X[i_start:i_end])
            cap = np.inf if i_start == 0 else self.acq_cond[i_start - 1]
            a_cond = np.clip(self.f(self.y[i_start:i_end], s_cond), None, cap)
            j = np.argsort(-a_cond)
            if a_cond[j[0]] == -np.inf:
                self.acq_cond[i_start:i_end] = -np.inf
                return
            src = i_start + j
            for arr in (self.X, self.y, self.sigma, self.acq):
                arr[i_start:i_end] = arr[src]
            self.acq_cond[i_start:i_end] = a_cond[j]
            i_start += 1

    def filled(self):
        """``copy(drop_empty=True)`` (:1575-1596): number of leading finite slots."""
        empt = np.flatnonzero(self.acq_cond[:-1] == -np.inf)
        return empt[0] if len(empt) else len(self)


# ----------------------------------------------------------------------------
# a13: NORA.multi_add with an injected candidate pool
# ----------------------------------------------------------------------------
def nora_multi_add(gpr, X_mc, n_points, zeta=None, already_proposed=None,
                   method="single sort acq", return_all=False):
    """gpry/gp_acquisition.py:971-1108, single process, pool ``X_mc`` given.

    Returns ``(X_pool, y_pool, acq_pool)`` and, with ``return_all``, a dict of the
    intermediate arrays (y, sigma, acq for every candidate; final pool state).
    """
    zeta = auto_zeta(gpr.d) if zeta is None else zeta
    y_mc, s_mc = gpr.predict(X_mc, return_std=True)  # mpi.compute_y_parallel, 1 rank
    keep = np.ones(len(X_mc), dtype=bool)
    if already_proposed is not None and len(already_proposed):
        for row in already_proposed:  # :1037-1047
            hit = np.flatnonzero(np.all(np.isin(X_mc, row, assume_unique=True), axis=1))
            if hit.size:
                keep[hit[0]] = False
    X_use, y_use, s_use = X_mc[keep], y_mc[keep], s_mc[keep]

    def f(mu, std):
        return logexp_f(mu, std, gpr.y_max, gpr.noise_level, zeta)

    acq = f(y_use, s_use)
    pool = OracleRankedPool(n_points, gpr, f)
    pool.add(X_use, y_use, s_use, acq, method=method)
    m = min(pool.filled(), n_points)
    X_pool, y_pool = pool.X[:m].copy(), pool.y[:m].copy()
    acq_pool = f(y_pool, pool.sigma[:m])
    if return_all:
        return X_pool, y_pool, acq_pool, dict(
            y=y_mc, sigma=s_mc, acq=acq, keep=keep, acq_cond=pool.acq_cond.copy(),
            cache_counter=pool.cache_counter, n_examined=pool.n_examined)
    return X_pool, y_pool, acq_pool


# ----------------------------------------------------------------------------
# synthetic workloads of BASELINE.md section 3 / SURVEY.md section 8d
# ----------------------------------------------------------------------------
def synthetic_problem(N, d, M, seed_train=0, seed_cand=1):
    """Correlated-Gaussian log-posterior training set + candidate pool."""
    rng = np.random.default_rng(seed_train)
    A = rng.standard_normal((d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    bounds = np.array([[-5.0, 5.0]] * d)
    Lc = np.linalg.cholesky(Sigma)
    X = np.clip(rng.standard_normal((N, d)) @ Lc.T, -5, 5)
    Sinv = np.linalg.inv(Sigma)
    y = -0.5 * np.einsum("ni,ij,nj->n", X, Sinv, X)
    rng_c = np.random.default_rng(seed_cand)
    Xc = np.clip(rng_c.standard_normal((M, d)) @ (math.sqrt(1.5) * Lc).T, -5, 5)
    return bounds, X, y, Xc


def synthetic_like_goldens(N, d, M, seed):
    """Same generator as tools/make_goldens.py:gauss_problem (one RNG stream)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    Lc = np.linalg.cholesky(Sigma)
    bounds = np.array([[-5.0, 5.0]] * d)
    X = np.clip(rng.standard_normal((N, d)) @ Lc.T, -5, 5)
    y = -0.5 * np.einsum("ni,ij,nj->n", X, np.linalg.inv(Sigma), X)
    Xc = np.clip(rng.standard_normal((M, d)) @ (np.sqrt(1.5) * Lc).T, -5, 5)
    return bounds, X, y, Xc


def curved_degeneracy(X, a=10., b=0.45, c=4., d=20.):
    """tests/model_generator.py:134 (2-d only); bounds at :124."""
    x0, x1 = X[..., 0], X[..., 1]
    return -(a * (b - x0)) ** 2. / c - (d * (x1 / c - x0 ** 4.)) ** 2.


CURVED_BOUNDS = np.array([[-0.5, 1.5], [-0.5, 2.]])
