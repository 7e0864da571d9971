"""Acquisition functions evaluated on (mean, std) pairs.

Only the value path of ``LogExp`` is on the hot path (gpry/acquisition_functions.py:
``BaseLogExp.__call__`` :936-1009, ``LogExp.f`` :1068-1074, ``auto_zeta`` :933-934).
In the NORA sweep the same formula runs fused on the device (``sweep_finish_kernel``);
the host version below is used for the handful of pool points that the ranking
re-evaluates with conditioned standard deviations.
"""
import warnings
from collections.abc import Iterable

import numpy as np


def is_acquisition_function(obj):
    return isinstance(obj, AcquisitionFunction)


def builtin_names():
    return ["LogExp"]


class AcquisitionFunction:
    hasgradient = False

    def check_X(self, X):
        return np.atleast_2d(X)


def _quiet_predict(gp, X, with_gradients):
    """``gp.predict`` with std (and both x-gradients); the GP's warnings about clipped variances are not the caller's."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if with_gradients:
            return gp.predict(X, return_std=True, return_mean_grad=True, return_std_grad=True)
        return gp.predict(X, return_std=True) + (None, None)


class LogExp(AcquisitionFunction):
    r"""``log A = 2 zeta (mu - baseline) + log sqrt(max(sigma^2 - sigma_n^2, 0))``."""

    def __init__(self, zeta=None, sigma_n=None, fixed=False, dimension=None, zeta_scaling=0.85,
                 linear=True):
        if zeta is None and dimension is None:
            raise ValueError("We need the dimensionality of the problem to guess an "
                             "appropriate zeta value.")
        self.zeta = self.auto_zeta(dimension, scaling=zeta_scaling) if zeta is None else zeta
        self.sigma_n, self.fixed = sigma_n, fixed
        self.hasgradient = True      # x-gradient branch of __call__ (gpry/acquisition_functions.py:993-1007)

    @staticmethod
    def auto_zeta(dimension, scaling=0.85):
        return dimension ** (-scaling)

    @staticmethod
    def f(mu, std, baseline, noise_level, zeta):
        """Same operation order as the reference (square, subtract, clip at 0, sqrt, log);
        ``np.maximum(x, 0.)`` is ``np.clip(x, 0., None)`` without its Python-level overhead."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return (2 * zeta * (mu - baseline) +
                    np.log(np.sqrt(np.maximum(std ** 2. - noise_level ** 2., 0.))))

    def _noise(self, gp):
        """(scalar noise of the value, noise as the gradient uses it): the reference averages a per-point
        ``gp.noise_level`` for the value but divides by the raw one in the gradient (:993-1007)."""
        if self.sigma_n is not None:
            return self.sigma_n, self.sigma_n
        raw = gp.noise_level
        return (np.mean(raw) if isinstance(raw, Iterable) else raw), raw

    def __call__(self, X, gp, eval_gradient=False):
        """Value (and x-gradient, gpry/acquisition_functions.py:937-1009) at ``X``: -inf (gradient +inf) where the
        mean is not finite or the variance does not exceed the noise."""
        mu, std, mu_grad, std_grad = _quiet_predict(gp, self.check_X(X), eval_gradient)
        noise, sigma_n = self._noise(gp)
        informative = (std ** 2 - noise ** 2. > 0) & np.isfinite(mu)
        values = np.full_like(std, -np.inf)
        if informative.any():
            values[informative] = self.f(mu[informative], std[informative], gp.y_max, noise, self.zeta)
        if not eval_gradient:
            return values
        # std_grad / (std - sigma_n) + 2 zeta mu_grad
        if np.ndim(std_grad) > 1:                       # one row per point
            mu_grad, std_grad = np.asarray(mu_grad), np.asarray(std_grad)
            grad = np.full_like(std_grad, np.inf)
            if informative.any():
                grad[informative] = (std_grad[informative] / (std[informative] - sigma_n)
                                     + 2 * self.zeta * mu_grad[informative])
            return values, grad
        if std[0] > sigma_n:                            # a single point, gradients as vectors
            return values, std_grad / (std[0] - sigma_n) + 2 * self.zeta * mu_grad
        return values, np.full_like(std_grad, np.inf)

    def value_and_grad_rows(self, X, gp):
        """``__call__(x, gp, eval_gradient=True)`` for every row of ``X`` from ONE batched posterior evaluation
        (``gp.predict_with_gradients``): ``(values (m,), grads (m, d))`` with the single-point conventions row by row --
        including the reference's two different tests for a usable gradient (:993-1007: the informative-point mask when
        the GP hands out 2-d gradient arrays, i.e. has a classifier; ``std > sigma_n`` otherwise).  For the optimiser runs
        that are stepped side by side (``gpry_amd.lockstep``); scalar noise only."""
        X = self.check_X(X)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mu, std, mu_grad, std_grad = gp.predict_with_gradients(X)
        noise, sigma_n = self._noise(gp)
        if isinstance(sigma_n, Iterable):
            raise ValueError("value_and_grad_rows needs a scalar noise level")
        informative = (std ** 2 - noise ** 2. > 0) & np.isfinite(mu)
        values = np.full_like(std, -np.inf)
        if informative.any():
            values[informative] = self.f(mu[informative], std[informative], gp.y_max, noise, self.zeta)
        usable = informative if getattr(gp, "infinities_classifier", None) is not None else std > sigma_n
        grads = np.full_like(std_grad, np.inf)
        if usable.any():
            grads[usable] = std_grad[usable] / (std[usable] - sigma_n)[:, None] + 2 * self.zeta * mu_grad[usable]
        return values, grads

    def __repr__(self):
        return f"LogExp(zeta={self.zeta:.3f})"
