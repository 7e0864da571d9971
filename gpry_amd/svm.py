"""Finite / infinite classifier that gates GP predictions.

Same role and interface as ``gpry/svm.py`` (``fit`` :227, ``_is_finite_raw`` :273-295,
``is_finite`` :297, ``predict`` :308-347): an RBF support-vector classifier trained on
"y above (max - threshold)" labels.  Training stays on the host (libsvm through scikit-learn);
its decision function -- an RBF expansion over the support vectors -- is exported with
``device_params`` so that the NORA sweep evaluates it on the GPU for its 1e5-1e6 candidates
(``gpry_set_gates``), and ``GaussianProcessRegressor.predict`` lets the device apply it to its points as well
(option ``predict_gates``: the one-point calls of the samplers would otherwise spend 50-100 us in libsvm per call).

Unlike the reference this is not a subclass of ``sklearn.svm.SVC``: it owns one (``_svc``) only while the labels are
mixed, which are the only training sets libsvm has anything to learn from.  The public attributes GPry reads
(``y_finite``, ``all_finite``, ``at_least_one_finite``, ``diff_threshold``, ``abs_threshold``, ``n``, ``d``) are kept.
"""
import warnings

import numpy as np

from gpry_amd.tools import check_random_state

_UNTRAINED = "The SVM has not been trained yet."


class SVM:
    def __init__(self, C=1e7, kernel="rbf", gamma="scale", tol=1e-3, random_state=None, **svc_kwargs):
        self._svc_args = dict(C=C, kernel=kernel, gamma=gamma, tol=tol, **svc_kwargs)
        self.random_state = check_random_state(random_state, convert_to_random_state=True)
        self._svc = None
        self._fit_count = 0        # bumped by every fit: what the device-side copy of the decision function is keyed by
        self.X_train = self.y_train = self.y_finite = None
        self.diff_threshold = self._max_y = None
        self.at_least_one_finite = self.all_finite = False

    # -- the training set -------------------------------------------------------------------------
    n = property(lambda self: 0 if self.y_train is None else len(self.y_train), doc="number of training points")

    @property
    def d(self):
        if self.X_train is None:
            raise ValueError("You need to add some data before determining its dimension.")
        return self.X_train.shape[1]

    @property
    def abs_threshold(self):
        """Lowest target that still counts as finite."""
        return self._max_y - self.diff_threshold

    # -- labels -----------------------------------------------------------------------------------
    @staticmethod
    def _is_finite_raw(y, diff_threshold, max_y=None):
        """Threshold test (not a prediction).  ``isfinite`` matters for y = +inf with an infinite threshold and
        for NaN."""
        top = np.max(y) if max_y is None else max_y
        return np.greater_equal(y, top - diff_threshold) & np.isfinite(y)

    def is_finite(self, y):
        if self.y_train is None:
            raise ValueError("Cannot do anything: the SVM has not been trained yet!")
        return self._is_finite_raw(y, self.diff_threshold, self._max_y)

    # -- training ---------------------------------------------------------------------------------
    def fit(self, X, y, diff_threshold):
        """Label the targets against ``max(y) - diff_threshold`` and train libsvm if both labels occur; returns the
        labels.  A training set of -inf only keeps the previous threshold (as the reference does)."""
        self._fit_count = getattr(self, "_fit_count", 0) + 1
        self.X_train, self.y_train = np.array(X, copy=True), np.array(y, copy=True)
        self.at_least_one_finite = not np.all(self.y_train == -np.inf)
        if not self.at_least_one_finite:
            self.y_finite = np.zeros(len(X), dtype=bool)
            return self.y_finite
        self.diff_threshold, self._max_y = diff_threshold, max(self.y_train)
        self.y_finite = self._is_finite_raw(self.y_train, diff_threshold, max_y=self._max_y)
        self.all_finite = bool(self.y_finite.all())
        if not self.all_finite:
            from sklearn.svm import SVC
            self._svc = SVC(random_state=self.random_state, **self._svc_args).fit(self.X_train, self.y_finite)
        return self.y_finite

    def _has_decision_function(self):
        return self.y_train is not None and self.at_least_one_finite and not self.all_finite and self._svc is not None

    def device_params(self):
        """``(support_vectors, dual_coef, gamma, intercept, positive_is_finite)`` of the fitted
        two-class RBF SVC, with ``predict(x) == (sum_i coef_i exp(-gamma |x - sv_i|^2) + intercept
        > 0) == positive_is_finite``; ``None`` when there is nothing to evaluate on the device
        (not trained, all points finite or none, other kernel)."""
        svc = self._svc
        if not self._has_decision_function() or svc.kernel != "rbf" or len(svc.classes_) != 2:
            return None
        as_f64 = lambda a: np.ascontiguousarray(a, dtype=float)     # noqa: E731
        return (as_f64(svc.support_vectors_), as_f64(svc.dual_coef_[0]), float(svc._gamma),
                float(svc.intercept_[0]), bool(svc.classes_[1]))

    # -- prediction -------------------------------------------------------------------------------
    def predict(self, X, validate=True):
        """True where a finite target is expected.  (``validate`` is accepted for the reference's signature; the
        input always goes through ``atleast_2d``.)"""
        if self.y_train is None:
            raise ValueError(_UNTRAINED)
        X = np.atleast_2d(X)
        if self.all_finite or not self.at_least_one_finite:
            if not self.all_finite:
                warnings.warn("Only -inf points added to the classifier so far. Returning False unconditionally.")
            return np.full(len(X), bool(self.all_finite))
        return self._svc.predict(X).astype(bool)
