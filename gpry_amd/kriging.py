"""Kriging-believer conditioning without copies or refactorisation.

The reference builds every conditioned model of ``RankedPool`` by ``deepcopy(gpr)`` +
``append_to_data(pool_X, pool_y, fit_gpr=False, fit_classifier=False)``, i.e. a full
kernel rebuild + Cholesky + inverse, O(N^3) (gpry/gp_acquisition.py:1522-1555 ->
gpry/gpr.py:1015-1017).  Only ``predict_std`` of those models is ever used
(gp_acquisition.py:1356,1481,1631), and the factor of the augmented matrix

    [[K, k_p], [k_p^T, k_pp + alpha]]  =  [[L, 0], [u^T, l]] [[L, 0], [u^T, l]]^T ,  u = L^-1 k_p

extends the base factor by one *border* row per appended point.  With u(x) = V k*(x)
kept on the device for every candidate that is looked at (``gpry_kb_register``), the
conditioned variance is

    var_t(x) = C - |u(x)|^2 - sum_{s<=t} w_s(x)^2,
    w_s(x) = ( k(x, p_s) - u(p_s).u(x) - sum_{r<s} c_sr w_r(x) ) / l_s

where the only O(N) work per appended point is one device GEMV (``gpry_kb_gram``).
This equals the reference's refactorised model up to rounding.
"""
import copy

import numpy as np

from gpry_amd import _lib


class KBSession:
    """Candidates registered on the device for one (training set, theta) factor."""

    def __init__(self, gpr):
        self.gpr = gpr
        self.dev = gpr.device
        gpr._push_affine()
        self.dev.kb_reset()
        d = gpr.d
        self.X = np.empty((0, d))
        self.var0 = np.empty(0)          # C - |u|^2 in transformed units (unclamped)
        self.masked = np.empty(0, bool)  # classifier says "infinite": std is 0
        self._index = {}
        self._gram = {}
        self._last_cond = None           # most recent conditioned model (prefix re-use)
        self.C = float(np.exp(gpr._device_theta()[1][0]))
        self.std_y = gpr._y_affine()[1]

    @property
    def n(self):
        return len(self.var0)

    def register(self, X):
        """Session indices of the rows of ``X`` (new rows are sent to the device)."""
        X = np.ascontiguousarray(np.atleast_2d(X), dtype=float)
        idx = np.empty(len(X), dtype=np.int64)
        new_rows, new_pos = [], []
        for i, row in enumerate(X):
            key = row.tobytes()
            j = self._index.get(key)
            if j is None:
                j = self.n + len(new_rows)
                self._index[key] = j
                new_rows.append(row)
                new_pos.append(i)
            idx[i] = j
        if new_rows:
            Xn = np.array(new_rows)
            first, var0 = self.dev.kb_register(Xn)
            if first != self.n:
                raise _lib.GpryHipError("KB session out of sync with the device")
            self.X = np.vstack([self.X, Xn])
            self.var0 = np.append(self.var0, var0)
            clf = self.gpr.infinities_classifier
            if clf is None:
                m = np.zeros(len(Xn), bool)
            else:
                Xn_ = np.ascontiguousarray(self.gpr.preprocessing_X.transform(Xn))
                m = np.logical_not(clf.predict(Xn_, validate=False))
            self.masked = np.append(self.masked, m)
        return idx

    def gram(self, p):
        """(u(p).u(x), k(p, x)) for all registered x; recomputed when the session has grown."""
        g = self._gram.get(p)
        if g is None or len(g[0]) < self.n:
            g = self.dev.kb_gram(p, self.n)
            self._gram[p] = g
        return g

    def noise_alpha(self):
        """alpha of appended points: they inherit the scalar noise level (gpry/gpr.py:711-715);
        per-point noise arrays make the reference raise (gpry/gpr.py:775-779)."""
        nl = self.gpr.noise_level
        if np.iterable(nl):
            raise ValueError("Need to pass non-null noise_level (scalar or array) because concrete "
                             "values were given earlier for the training points.")
        return float(self.gpr.preprocessing_y.transform_scale(nl)) ** 2

    def conditioned(self, Xp, yp):
        gpr = self.gpr
        clf = gpr.infinities_classifier
        if clf is not None:
            # the reference re-classifies old and new targets together with the lies
            y_all = np.append(gpr.y_train_all, yp)
            thr = gpr._diff_threshold_if_keep_n_finite(y_all, gpr.keep_min_finite, gpr._diff_threshold)
            fin = clf._is_finite_raw(y_all, thr)
            old_fin = clf._is_finite_raw(gpr.y_train_all, gpr._diff_threshold_if_keep_n_finite(
                gpr.y_train_all, gpr.keep_min_finite, gpr._diff_threshold))
            if not np.array_equal(fin[:len(old_fin)], old_fin):
                # the lies moved the threshold across old points: training set changes, so
                # build the model the long way (exactly what the reference does)
                c = copy.deepcopy(gpr)
                c.append_to_data(Xp, yp, fit_gpr=False, fit_classifier=False)
                return c
            Xp = Xp[fin[len(old_fin):]]
        return ConditionedGPR(self, self.register(Xp))


class ConditionedGPR:
    """``predict_std`` of the base GP augmented by session candidates ``idx`` (in order)."""

    def __init__(self, session, idx):
        self.s = session
        self.idx = [int(i) for i in idx]
        t_n = len(self.idx)
        self.c = np.zeros((t_n, t_n))
        self.l = np.zeros(t_n)
        alpha = session.noise_alpha() if t_n else 0.0
        self._rows = []
        # RankedPool asks for the models of pool[:1], pool[:2], ... in turn: border rows (and candidate
        # weights) that the previous model already holds for a common prefix are copied, not recomputed
        prev, t0 = session._last_cond, 0
        if prev is not None and 0 < len(prev.idx) <= t_n and prev.idx == self.idx[:len(prev.idx)]:
            t0 = len(prev.idx)
            self.c[:t0, :t0] = prev.c
            self.l[:t0] = prev.l
            self._rows = list(prev._rows)
        for t, p in enumerate(self.idx):
            if t < t0:
                continue
            G, kv = session.gram(p)
            self._rows.append((G, kv))
            acc = 0.0
            for r in range(t):
                q = self.idx[r]
                c_tr = (kv[q] - G[q] - np.dot(self.c[t, :r], self.c[r, :r])) / self.l[r]
                self.c[t, r] = c_tr
                acc += c_tr * c_tr
            l2 = (kv[p] + alpha) - G[p] - acc
            if not l2 > 0:
                raise np.linalg.LinAlgError(
                    f"{t + 1}-th appended point makes the augmented kernel matrix non positive "
                    "definite. Try gradually increasing the 'noise_level' parameter of your "
                    "GaussianProcessRegressor estimator.")
            self.l[t] = np.sqrt(l2)
        self._W = np.zeros((t_n, 0))
        if t0 and prev._W.shape[1]:
            have = prev._W.shape[1]
            W = np.zeros((t_n, have))
            W[:t0] = prev._W
            for t in range(t0, t_n):
                G, kv = session.gram(self.idx[t])
                W[t] = (kv[:have] - G[:have] - self.c[t, :t] @ W[:t]) / self.l[t]
            self._W = W
        session._last_cond = self
        self.n_eval = 0

    def _weights(self, n):
        """w_t(x) for the first ``n`` registered candidates (extended lazily)."""
        have = self._W.shape[1]
        if n > have:
            t_n = len(self.idx)
            W = np.zeros((t_n, n))
            W[:, :have] = self._W
            for t, p in enumerate(self.idx):
                G, kv = self.s.gram(p)
                w = kv[have:n] - G[have:n]
                if t:
                    w = w - self.c[t, :t] @ W[:t, have:n]
                W[t, have:n] = w / self.l[t]
            self._W = W
        return self._W

    def predict_std(self, X, validate=True):
        idx = self.s.register(X)
        self.n_eval += len(idx)
        W = self._weights(self.s.n)
        var = self.s.var0[idx] - np.sum(W[:, idx] ** 2, axis=0)
        var[var < 0] = 0.0                       # gpry/gpr.py:1337-1342
        std = np.sqrt(var) * self.s.std_y
        std[self.s.masked[idx]] = 0.0            # gpry/gpr.py:1314,1350
        return std
