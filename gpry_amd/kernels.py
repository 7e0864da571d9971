"""Kernel specification objects for the device GP.

They carry the hyperparameters, their (log) bounds and the operator structure with the
attribute surface of ``gpry/kernels.py`` (``Hyperparameter`` :26, ``Kernel.bounds``
:157-191, ``RBF`` :213, ``Matern`` :281, ``ConstantKernel`` :601, ``Product`` :681) and
of the scikit-learn kernels those derive from (``theta`` layout: sklearn:kernels.py:734-747).
All arithmetic -- ``k(X, Y)``, ``diag`` and the theta-gradient traces used by the
marginal likelihood -- runs on the GPU through ``gpry_amd._lib``; only
``ConstantKernel * RBF`` and ``ConstantKernel * Matern(nu in {0.5, 1.5, 2.5})``, the
kernels the reference's auto-constructor builds (gpry/gpr.py:343-363), are supported.
"""
import copy
import warnings
from collections import namedtuple

import numpy as np


class Hyperparameter(namedtuple("Hyperparameter", ("name", "value_type", "bounds", "max_length",
                                                   "n_elements", "fixed", "dynamic"))):
    """Specification of one kernel hyperparameter (gpry/kernels.py:26-115)."""
    __slots__ = ()

    def __new__(cls, name, value_type, bounds, max_length=None, n_elements=1, fixed=None,
                dynamic=None):
        if not isinstance(bounds, str):
            bounds = np.atleast_2d(bounds)
            if n_elements > 1:
                if bounds.shape[0] == 1:
                    bounds = np.repeat(bounds, n_elements, 0)
                elif bounds.shape[0] != n_elements:
                    raise ValueError(f"Bounds on {name} should have either 1 or {n_elements} "
                                     f"dimensions. Given are {bounds.shape[0]}")
        elif bounds not in ("fixed", "dynamic"):
            raise ValueError(f"bounds must be numeric, 'fixed' or 'dynamic'; got {bounds!r}")
        if fixed is None:
            fixed = isinstance(bounds, str) and bounds == "fixed"
        if dynamic is None:
            dynamic = isinstance(bounds, str) and bounds == "dynamic"
        return super().__new__(cls, name, value_type, bounds, max_length, n_elements, fixed, dynamic)


class Kernel:
    """Common behaviour: theta <-> parameter mapping, bounds, operators."""
    requires_vector_input = True

    # -- parameters ---------------------------------------------------------------------
    def get_params(self, deep=True):
        raise NotImplementedError

    @property
    def hyperparameters(self):
        names = type(self).__dict__.get("_hp_names")
        if names is None:          # the attribute names are a property of the class: dir() once, not per access
            names = [a for a in dir(type(self)) if a.startswith("hyperparameter_")]
            type(self)._hp_names = names
        return [getattr(self, a) for a in names]

    @property
    def n_dims(self):
        return self.theta.shape[0]

    @property
    def theta(self):
        vals = []
        params = self.get_params()
        for h in self.hyperparameters:
            if not h.fixed:
                vals.append(np.atleast_1d(params[h.name]).astype(float))
        return np.log(np.hstack(vals)) if vals else np.array([])

    @theta.setter
    def theta(self, theta):
        theta = np.asarray(theta, dtype=float)
        i = 0
        for h in self.hyperparameters:
            if h.fixed:
                continue
            if h.n_elements > 1:
                self._set_param(h.name, np.exp(theta[i:i + h.n_elements]))
                i += h.n_elements
            else:
                self._set_param(h.name, float(np.exp(theta[i])))
                i += 1
        if i != len(theta):
            raise ValueError(f"theta has not the correct number of entries. Should be {i}; "
                             f"given are {len(theta)}")

    def _set_param(self, name, value):
        setattr(self, name, value)

    @property
    def bounds(self):
        """Log-transformed bounds of theta, with the reference's "dynamic" rule
        (gpry/kernels.py:157-191)."""
        rows = []
        params = self.get_params()
        for h in self.hyperparameters:
            if h.fixed:
                continue
            if h.dynamic:
                vals = np.atleast_1d(params[h.name])
                for t, v in enumerate(vals):
                    ref = v if (h.max_length is None or h.max_length[t] is None) else h.max_length[t]
                    rows.append([ref * 1e-3, ref * 100.0])
            else:
                rows.append(h.bounds)
        return np.log(np.vstack(rows)) if rows else np.array([])

    def clone_with_theta(self, theta):
        c = copy.deepcopy(self)
        c.theta = theta
        return c

    # -- operators ----------------------------------------------------------------------
    def __mul__(self, b):
        return Product(self, b if isinstance(b, Kernel) else ConstantKernel(b))

    def __rmul__(self, b):
        return Product(b if isinstance(b, Kernel) else ConstantKernel(b), self)

    # -- evaluation (device) --------------------------------------------------------------
    def device_spec(self, d):
        """(kernel_id, full theta [log C, log l_1..l_d]) for libgpry_hip.so."""
        raise NotImplementedError(
            f"{self!r}: only ConstantKernel * RBF / Matern(nu=0.5|1.5|2.5) run on the device")

    def __call__(self, X, Y=None, eval_gradient=False):
        from gpry_amd.gpr import _scratch_device
        if eval_gradient:
            raise NotImplementedError(
                "The (N, N, n_theta) gradient tensor is never materialised on the device; use "
                "GaussianProcessRegressor.log_marginal_likelihood(theta, eval_gradient=True).")
        X = np.atleast_2d(np.asarray(X, dtype=float))
        kid, th = self.device_spec(X.shape[1])
        dev = _scratch_device()
        if Y is None:
            dev.set_train(X, np.zeros(len(X)), np.zeros(len(X)))
            dev.set_theta(kid, th)
            return dev.kernel_train(add_alpha=False)
        Y = np.atleast_2d(np.asarray(Y, dtype=float))
        dev.set_train(Y, np.zeros(len(Y)), np.zeros(len(Y)))
        dev.set_theta(kid, th)
        return dev.kernel_cross(X)

    def gradient_x(self, x, X_train):
        """d k(x, X_train) / d x, shape (n_train, d), for one point ``x`` (gpry/kernels.py:193-210,
        RBF :257-278, Matern :326-432, product rule :687-699), evaluated on the device.
        (The reference's own Matern(nu=0.5) path raises through ``Product.gradient_x`` because of
        a shape slip at :355-359; here it returns the gradient that code describes.)"""
        from gpry_amd.gpr import _scratch_device
        X_train = np.atleast_2d(np.asarray(X_train, dtype=float))
        x = np.asarray(x, dtype=float).reshape(-1)
        kid, th = self.device_spec(X_train.shape[1])
        dev = _scratch_device()
        dev.set_train(X_train, np.zeros(len(X_train)), np.zeros(len(X_train)))
        dev.set_theta(kid, th)
        dev.set_affine()
        return dev.predict_grad(x, want_kinv=False, want_kgrad=True, want_mean=False)[2]

    def diag(self, X):
        return np.diag(self(X))


class ConstantKernel(Kernel):
    def __init__(self, constant_value=1.0, constant_value_bounds=(1e-5, 1e5)):
        self.constant_value = constant_value
        self.constant_value_bounds = constant_value_bounds

    def get_params(self, deep=True):
        return {"constant_value": self.constant_value,
                "constant_value_bounds": self.constant_value_bounds}

    @property
    def hyperparameter_constant_value(self):
        return Hyperparameter("constant_value", "numeric", self.constant_value_bounds, None)

    def diag(self, X):
        return np.full(np.atleast_2d(X).shape[0], self.constant_value, dtype=float)

    def __repr__(self):
        return "{0:.3g}**2".format(np.sqrt(self.constant_value))


class _Stationary(Kernel):
    """Shared part of RBF and Matern: anisotropic length scales, "dynamic" bounds."""

    def __init__(self, length_scale=1.0, length_scale_bounds=(1e-5, 1e5), prior_bounds=None):
        self.length_scale = length_scale
        self.length_scale_bounds = length_scale_bounds
        self.prior_bounds = prior_bounds
        self.max_length = None
        if isinstance(length_scale_bounds, str) and length_scale_bounds == "dynamic":
            if prior_bounds is None:
                raise TypeError(f"Prior bounds are required for the {type(self).__name__} kernel "
                                "if its hyperparameter bounds are set to 'dynamic'.")
            if not np.iterable(prior_bounds):
                raise TypeError("prior_bounds needs to be an iterable.")
            pb = np.asarray(prior_bounds)
            if not self.anisotropic and pb.shape[0] > 1:
                warnings.warn("Isotropic kernel with 'dynamic' bounds in more than one dimension.")
            self.max_length = pb[:, 1] - pb[:, 0]

    @property
    def anisotropic(self):
        ls = self.length_scale
        if isinstance(ls, np.ndarray):          # (np.iterable costs 5 us; the objective asks three times per evaluation)
            return ls.ndim > 0 and ls.shape[0] > 1
        return np.iterable(ls) and len(ls) > 1

    @property
    def hyperparameter_length_scale(self):
        n = len(self.length_scale) if self.anisotropic else 1
        return Hyperparameter("length_scale", "numeric", self.length_scale_bounds, self.max_length, n)

    def _set_param(self, name, value):
        if name == "length_scale" and not self.anisotropic:
            value = float(np.atleast_1d(value)[0])
            if np.iterable(self.length_scale):  # keep a length-1 container as given
                value = type(self.length_scale)([value]) if not isinstance(
                    self.length_scale, np.ndarray) else np.array([value])
        setattr(self, name, value)

    def diag(self, X):
        return np.ones(np.atleast_2d(X).shape[0])

    def _fmt_ls(self):
        if self.anisotropic:
            return "[" + ", ".join("{0:.3g}".format(v) for v in self.length_scale) + "]"
        return "{0:.3g}".format(np.ravel(self.length_scale)[0])


class RBF(_Stationary):
    def get_params(self, deep=True):
        return {"length_scale": self.length_scale, "length_scale_bounds": self.length_scale_bounds,
                "prior_bounds": self.prior_bounds}

    kernel_id = 0

    def __repr__(self):
        return f"RBF(length_scale={self._fmt_ls()})"


class Matern(_Stationary):
    def __init__(self, length_scale=1.0, length_scale_bounds=(1e-5, 1e5), nu=1.5, prior_bounds=None):
        super().__init__(length_scale, length_scale_bounds, prior_bounds)
        self.nu = nu

    def get_params(self, deep=True):
        return {"length_scale": self.length_scale, "length_scale_bounds": self.length_scale_bounds,
                "nu": self.nu, "prior_bounds": self.prior_bounds}

    @property
    def kernel_id(self):
        try:
            return {0.5: 1, 1.5: 2, 2.5: 3}[float(self.nu)]
        except KeyError:
            raise NotImplementedError(
                f"Matern(nu={self.nu}) is not available on the device (nu must be 0.5, 1.5 or 2.5)")

    def __repr__(self):
        return f"Matern(length_scale={self._fmt_ls()}, nu={self.nu:.3g})"


class Product(Kernel):
    def __init__(self, k1, k2):
        self.k1, self.k2 = k1, k2

    def get_params(self, deep=True):
        p = {"k1": self.k1, "k2": self.k2}
        if deep:
            p.update({"k1__" + k: v for k, v in self.k1.get_params().items()})
            p.update({"k2__" + k: v for k, v in self.k2.get_params().items()})
        return p

    @property
    def hyperparameters(self):
        out = []
        for pre, k in (("k1__", self.k1), ("k2__", self.k2)):
            for h in k.hyperparameters:
                out.append(Hyperparameter(pre + h.name, h.value_type, h.bounds, h.max_length,
                                          h.n_elements, h.fixed, h.dynamic))
        return out

    @property
    def theta(self):
        return np.append(self.k1.theta, self.k2.theta)

    @theta.setter
    def theta(self, theta):
        n1 = self.k1.n_dims
        self.k1.theta = theta[:n1]
        self.k2.theta = theta[n1:]

    @property
    def bounds(self):
        b1, b2 = self.k1.bounds, self.k2.bounds
        if b1.size == 0:
            return b2
        if b2.size == 0:
            return b1
        return np.vstack((b1, b2))

    def _parts(self):
        if isinstance(self.k1, ConstantKernel) and isinstance(self.k2, _Stationary):
            return self.k1, self.k2
        if isinstance(self.k2, ConstantKernel) and isinstance(self.k1, _Stationary):
            return self.k2, self.k1
        raise NotImplementedError(
            f"{self!r}: only ConstantKernel * RBF / Matern products run on the device")

    def device_spec(self, d):
        const, stat = self._parts()
        lsv = stat.length_scale
        if isinstance(lsv, np.ndarray) and lsv.shape == (d,) and lsv.dtype == np.float64:
            full = np.empty(d + 1)
            full[0] = const.constant_value
            full[1:] = lsv
            return stat.kernel_id, np.log(full)
        ls = np.broadcast_to(np.atleast_1d(np.asarray(stat.length_scale, dtype=float)), (d,)) \
            if np.size(stat.length_scale) in (1, d) else None
        if ls is None:
            raise ValueError(f"Anisotropic kernel must have the same number of dimensions as data "
                             f"({np.size(stat.length_scale)}!={d})")
        return stat.kernel_id, np.log(np.append(float(const.constant_value), ls))

    # -- the optimiser's objective sets theta and asks for the device vector once per evaluation: at small N the
    # generic path (hyperparameter lists rebuilt from dir(), namedtuples, three passes of exp / log) cost more than
    # the evaluation on the GPU (40-75 us against 30-60).  Same values, set through the same attributes.
    def _layout(self):
        const, stat = self._parts()
        cb, lb = const.constant_value_bounds, stat.length_scale_bounds
        n_c = 0 if (isinstance(cb, str) and cb == "fixed") else 1
        nls = len(stat.length_scale) if stat.anisotropic else 1
        n_l = 0 if (isinstance(lb, str) and lb == "fixed") else nls
        return const, stat, n_c, n_l, nls, const is self.k1

    def set_theta_and_full(self, theta, d):
        """``self.theta = theta`` followed by ``self.device_spec(d)`` in one pass."""
        const, stat, n_c, n_l, nls, const_first = self._layout()
        theta = np.asarray(theta, dtype=float)
        if theta.shape != (n_c + n_l,):
            raise ValueError(f"theta has not the correct number of entries. Should be {n_c + n_l}; "
                             f"given are {len(theta)}")
        e = np.exp(theta)
        ec, el = (e[:n_c], e[n_c:]) if const_first else (e[n_l:], e[:n_l])
        if n_c:
            const.constant_value = float(ec[0])
        if n_l:
            stat._set_param("length_scale", el if nls > 1 else float(el[0]))
        return self.device_spec(d)

    def grad_from_full_fast(self, grad_full, d):
        const, stat, n_c, n_l, nls, const_first = self._layout()
        if n_c == 1 and n_l == d and const_first:
            return np.array(grad_full[:d + 1], dtype=float)
        return self.grad_from_full(grad_full, d)

    def clone_with_theta(self, theta):
        # shallow copies of the two factors: the setter replaces the values it touches by new objects, bounds and
        # prior boxes are never modified in place (a deepcopy cost 45 us per objective evaluation)
        c = Product(copy.copy(self.k1), copy.copy(self.k2))
        c.set_theta_and_full(theta, np.size(self._parts()[1].length_scale))
        return c

    def theta_to_full(self, theta, d):
        """Map a (possibly reduced: fixed / isotropic) theta to the device's 1+d vector."""
        k = copy.deepcopy(self)
        k.theta = theta
        return k.device_spec(d)[1]

    def grad_from_full(self, grad_full, d):
        """Chain rule from d lml / d [log C, log l_1..l_d] to the free entries of theta."""
        const, stat = self._parts()
        g_const = [] if const.hyperparameter_constant_value.fixed else [grad_full[0]]
        h = stat.hyperparameter_length_scale
        if h.fixed:
            g_ls = []
        elif stat.anisotropic:
            g_ls = list(grad_full[1:1 + d])
        else:
            g_ls = [np.sum(grad_full[1:1 + d])]
        first, second = (g_const, g_ls) if const is self.k1 else (g_ls, g_const)
        return np.array(first + second, dtype=float)

    def diag(self, X):
        return self.k1.diag(X) * self.k2.diag(X)

    def __repr__(self):
        return f"{self.k1!r} * {self.k2!r}"


def clone(kernel):
    """Unfitted deep copy (what ``sklearn.base.clone`` does for the reference's kernels)."""
    return copy.deepcopy(kernel)
