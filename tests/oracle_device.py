"""Test double of ``gpry_amd._lib.Device``: same methods, numpy/scipy arithmetic from the oracle.

Lets the CPU suite (``-m "not gpu"``) drive the *host* side of the product (the
``GaussianProcessRegressor`` mirror, the restart farm, NORA sharding) end to end without a GPU.
Test infrastructure only: nothing under ``gpry_amd/`` imports it.
"""
import numpy as np

from oracle import gpry_oracle as orc
from gpry_amd._lib import CAND_DTYPE, MASK_CLASSIFIED_INF, MASK_OUTSIDE_TRUST


class OracleDevice:
    def __init__(self, device=0):
        self.device = int(device)
        self.N = self.d = 0
        self.kid, self.theta = orc.RBF, None
        self.lo = self.span = None
        self.y_mean, self.y_std, self.clip_hi = 0.0, 1.0, np.inf
        self.n_lml = self.n_factorize = 0
        self._factor = None

    # -- model state --------------------------------------------------------------------
    def set_train(self, X_, y_, alpha):
        self.X_ = np.array(X_, dtype=float)
        self.N, self.d = self.X_.shape
        self.y_ = np.array(y_, dtype=float)
        self.alpha = np.array(np.broadcast_to(alpha, (self.N,)), dtype=float)
        self._factor = None

    def set_theta(self, kernel_id, theta):
        self.kid, self.theta = int(kernel_id), np.array(theta, dtype=float)
        assert self.theta.shape == (self.d + 1,)
        self._factor = None

    def set_affine(self, x_lo=None, x_span=None, y_mean=0.0, y_std=1.0, clip_hi=np.inf):
        self.lo = None if x_lo is None else np.array(x_lo, dtype=float)
        self.span = None if x_span is None else np.array(x_span, dtype=float)
        self.y_mean, self.y_std, self.clip_hi = float(y_mean), float(y_std), float(clip_hi)

    def set_option(self, key, value):
        pass

    def set_gates(self, sv=None, coef=None, gamma=0.0, intercept=0.0, positive_is_finite=True,
                  trust_bounds=None):
        self.gates = None
        if (sv is not None and len(sv)) or trust_bounds is not None:
            self.gates = (None if sv is None else np.array(sv), None if coef is None else np.array(coef),
                          gamma, intercept, positive_is_finite,
                          None if trust_bounds is None else np.array(trust_bounds))

    def _gate_bits(self, X):
        sv, coef, gamma, intercept, pos, tb = self.gates
        bits = np.zeros(len(X), dtype=np.uint8)
        if tb is not None:
            bits[~np.all((X >= tb[:, 0]) & (X <= tb[:, 1]), axis=1)] |= MASK_OUTSIDE_TRUST
        if sv is not None:
            X_ = self._to_unit(X)
            d2 = ((X_[:, None, :] - sv[None, :, :]) ** 2).sum(-1)
            dec = np.exp(-gamma * d2).dot(coef) + intercept
            bits[(dec > 0) != pos] |= MASK_CLASSIFIED_INF
        return bits

    def sync(self):
        pass

    # -- factor / lml -------------------------------------------------------------------
    def factorize(self):
        self.n_factorize += 1
        K = orc.kernel_matrix(self.X_, self.theta, self.kid)
        K[np.diag_indices_from(K)] += self.alpha
        try:
            self._factor = orc.factorize(K, self.y_)
        except np.linalg.LinAlgError as e:
            self._factor = None
            msg = str(e)
            return int(msg.split("-th")[0]) if "-th" in msg else 1
        return 0

    def get_factor(self, want_L=True, want_V=True, want_alpha=True):
        L, V, a = self._factor
        return (L.copy() if want_L else None, V.copy() if want_V else None,
                a.copy() if want_alpha else None)

    def lml(self, theta, eval_gradient=False):
        self.n_lml += 1
        out = orc.log_marginal_likelihood(self.X_, self.y_, self.alpha, np.asarray(theta, float),
                                          self.kid, eval_gradient=eval_gradient)
        if eval_gradient:
            return float(out[0]), np.array(out[1]), int(not np.isfinite(out[0]))
        return float(out), int(not np.isfinite(out))

    lml_batch_max = 2048      # the device's "lml_batch" option: largest training set whose batch is ONE chain of launches

    def lml_batch(self, thetas, eval_gradient=True):
        rows = [self.lml(th, eval_gradient) for th in np.atleast_2d(thetas)]
        if eval_gradient:
            return np.array([r[0] for r in rows]), np.array([r[1] for r in rows]), np.array([r[2] for r in rows])
        return np.array([r[0] for r in rows]), np.array([r[1] for r in rows])

    # -- predict / sweep ----------------------------------------------------------------
    def _to_unit(self, X):
        X = np.atleast_2d(np.asarray(X, dtype=float))
        return X if self.lo is None else (X - self.lo) / self.span

    def predict(self, X, return_std=False, mask=None):
        X_ = self._to_unit(X)
        L, V, a = self._factor
        Kt = orc.kernel_matrix(X_, self.theta, self.kid, Y=self.X_)
        mean = np.minimum(Kt.dot(a) * self.y_std + self.y_mean, self.clip_hi)
        if mask is not None:
            mask = np.asarray(mask, dtype=np.uint8)
            mean[(mask & (MASK_CLASSIFIED_INF | MASK_OUTSIDE_TRUST)) != 0] = -np.inf
        if not return_std:
            return mean
        Mm = V.dot(Kt.T)
        var = orc.kernel_diag(X_, self.theta) - np.einsum("ji,ji->i", Mm, Mm)
        var[var < 0] = 0.0
        std = np.sqrt(var) * self.y_std
        if mask is not None:
            std[(mask & MASK_CLASSIFIED_INF) != 0] = 0.0
        return mean, std

    def predict_grad(self, x, want_kinv=True, want_kgrad=False, want_mean=True):
        x_ = self._to_unit(np.asarray(x, dtype=float)[None, :])[0]
        G = orc.kernel_gradient_x(x_, self.X_, self.theta, self.kid)
        mg = kg = None
        if want_mean or want_kinv:
            L, V, a = self._factor
            mg = G.T.dot(a)
            kg = np.zeros(self.d)
            if want_kinv:
                Kt = orc.kernel_matrix(x_[None, :], self.theta, self.kid, Y=self.X_)
                kg = np.dot(Kt, np.dot(V.T.dot(V), G))[0]
        return (mg, kg, G) if want_kgrad else (mg, kg)

    def predict_point(self, x, mask_bits=0, want_kinv=True):
        x = np.asarray(x, dtype=float)
        bits = int(mask_bits)
        if getattr(self, "gates", None) is not None and getattr(self, "applies_gates_in_predict", False):
            bits |= int(self._gate_bits(x[None, :])[0])
        mask = None if not bits else np.array([bits], dtype=np.uint8)
        mean, std = OracleDevice.predict(self, x[None, :], return_std=True, mask=mask)
        mg, kg = self.predict_grad(x, want_kinv=want_kinv)
        return float(mean[0]), float(std[0]), mg, kg, bits

    def predict_grad_batch(self, X, want_kinv=True):
        X = np.atleast_2d(np.asarray(X, dtype=float))
        mean, std = self.predict(X, return_std=True)
        both = [self.predict_grad(x, want_kinv=want_kinv) for x in X]
        return mean, std, np.array([b[0] for b in both]), np.array([b[1] for b in both])

    def append_rows(self, Xnew_, ynew_, alphanew):
        """Bordered update stands in as a refactorisation of the enlarged set (same numbers to rounding)."""
        X_ = np.vstack([self.X_, np.atleast_2d(Xnew_)])
        y_ = np.append(self.y_, ynew_)
        a = np.append(self.alpha, np.broadcast_to(alphanew, (len(np.atleast_1d(ynew_)),)))
        theta, kid = self.theta, self.kid
        self.set_train(X_, y_, a)
        self.theta, self.kid = theta, kid
        self.n_border = getattr(self, "n_border", 0) + 1
        return self.factorize()

    def sweep_logexp(self, X, zeta, baseline, sigma_n, mask=None, M=None, want=("y", "sigma", "acq")):
        if X is None:
            X = self._pool
        self._pool = np.array(X, dtype=float)
        if getattr(self, "gates", None) is not None and len(X):
            g = self._gate_bits(self._pool)
            mask = g if mask is None else (np.asarray(mask, dtype=np.uint8) | g)
        if len(X):
            y, s = self.predict(X, return_std=True, mask=mask)
        else:
            y, s = np.empty(0), np.empty(0)
        self.acq, self.y, self.s = orc.logexp_f(y, s, baseline, sigma_n, zeta), y, s
        self.sweep_epoch = getattr(self, "sweep_epoch", 0) + 1
        self.n_fetch = getattr(self, "n_fetch", 0)
        return {"y": y if "y" in want else None, "sigma": s if "sigma" in want else None,
                "acq": self.acq if "acq" in want else None, "n_nan": int(np.isnan(self.acq).sum())}

    def sweep_fetch(self, want=("y", "sigma")):
        self.n_fetch += 1
        src = {"y": self.y, "sigma": self.s, "acq": self.acq}
        return {k: (src[k].copy() if k in want else None) for k in src}

    # -- Kriging believer (u(x) = V k*(x) kept per registered candidate) -------------------
    def kb_reset(self):
        self._kb_X, self._kb_U = [], []

    def kb_register(self, X, want_var0=True):
        X_ = self._to_unit(X)
        L, V, a = self._factor
        Kt = orc.kernel_matrix(X_, self.theta, self.kid, Y=self.X_)
        U = V.dot(Kt.T).T
        first = len(self._kb_X)
        self._kb_X.extend(list(X_))
        self._kb_U.extend(list(U))
        var0 = orc.kernel_diag(X_, self.theta) - np.einsum("ij,ij->i", U, U)
        return first, (var0 if want_var0 else None)

    def kb_gram(self, p, n):
        assert n == len(self._kb_X)
        U = np.array(self._kb_U)
        Xk = np.array(self._kb_X)
        kv = orc.kernel_matrix(Xk[p:p + 1], self.theta, self.kid, Y=Xk)[0]
        return U.dot(U[p]), kv

    def sweep_topk(self, K, exclude=None):
        M = len(self.acq)
        ok = np.ones(M, bool)
        if exclude is not None and len(exclude):
            ok[np.asarray(exclude, dtype=int)] = False
        order = np.lexsort((-np.arange(M), -self.acq))
        order = order[ok[order]]
        sel = order[:K]
        top = np.zeros(len(sel), dtype=CAND_DTYPE)
        top["acq"], top["y"], top["sigma"], top["idx"] = self.acq[sel], self.y[sel], self.s[sel], sel
        bound = self.acq[order[K]] if len(order) > K else -np.inf
        return top, bound


def attach(gpr, device=None):
    """Give a host-side ``gpry_amd.gpr.GaussianProcessRegressor`` the oracle-backed device."""
    gpr._dev = device or OracleDevice()
    return gpr
