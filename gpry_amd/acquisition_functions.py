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


class LogExp(AcquisitionFunction):
    r"""``log A = 2 zeta (mu - baseline) + log sqrt(max(sigma^2 - sigma_n^2, 0))``."""

    def __init__(self, zeta=None, sigma_n=None, fixed=False, dimension=None, zeta_scaling=0.85,
                 linear=True):
        if zeta is None:
            if dimension is None:
                raise ValueError("We need the dimensionality of the problem to guess an "
                                 "appropriate zeta value.")
            zeta = self.auto_zeta(dimension, scaling=zeta_scaling)
        self.zeta = zeta
        self.sigma_n = sigma_n
        self.fixed = fixed
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

    def __call__(self, X, gp, eval_gradient=False):
        """Value (and x-gradient, gpry/acquisition_functions.py:937-1009) at ``X``."""
        X = self.check_X(X)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if eval_gradient:
                mu, std, mu_grad, std_grad = gp.predict(X, return_std=True, return_mean_grad=True,
                                                        return_std_grad=True)
            else:
                mu, std = gp.predict(X, return_std=True)
        if self.sigma_n is None:
            sigma_n = gp.noise_level
            noise = np.mean(sigma_n) if isinstance(sigma_n, Iterable) else sigma_n
        else:
            noise = sigma_n = self.sigma_n
        var = std ** 2 - noise ** 2.
        mask = (var > 0) & np.isfinite(mu)
        values = np.full_like(std, -np.inf)
        if np.any(mask):
            values[mask] = self.f(mu[mask], std[mask], gp.y_max, noise, self.zeta)
        if not eval_gradient:
            return values
        # the reference's gradient: std_grad / (std - sigma_n) + 2 zeta mu_grad, +inf where
        # std <= sigma_n (:993-1007; it uses the raw gp.noise_level here, as we do)
        if np.array(std_grad).ndim > 1:
            grad = np.zeros_like(std_grad)
            if np.any(mask):
                grad[mask] = np.array(std_grad)[mask] / (std[mask] - sigma_n) + \
                    2 * self.zeta * np.array(mu_grad)[mask]
            if np.any(~mask):
                grad[~mask] = np.ones_like(std_grad[~mask]) * np.inf
        elif std[0] > sigma_n:
            grad = std_grad / (std[0] - sigma_n) + 2 * self.zeta * mu_grad
        else:
            grad = np.ones_like(std_grad) * np.inf
        return values, grad

    def __repr__(self):
        return f"LogExp(zeta={self.zeta:.3f})"
