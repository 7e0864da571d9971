"""Finite / infinite classifier that gates GP predictions.

Same role and interface as ``gpry/svm.py`` (``fit`` :227, ``_is_finite_raw`` :273-295,
``is_finite`` :297, ``predict`` :308-347): an RBF support-vector classifier trained on
"y above (max - threshold)" labels.  Training stays on the host (libsvm through scikit-learn);
its decision function -- an RBF expansion over the support vectors -- is exported with
``device_params`` so that the NORA sweep evaluates it on the GPU for its 1e5-1e6 candidates
(``gpry_set_gates``), and ``GaussianProcessRegressor.predict`` lets the device apply it to its points as well
(option ``predict_gates``: the one-point calls of the samplers would otherwise spend 50-100 us in libsvm per call).
"""
import warnings

import numpy as np

from gpry_amd.tools import check_random_state


class SVM:
    def __init__(self, C=1e7, kernel="rbf", gamma="scale", tol=1e-3, random_state=None, **svc_kwargs):
        self._svc_args = dict(C=C, kernel=kernel, gamma=gamma, tol=tol, **svc_kwargs)
        self.random_state = check_random_state(random_state, convert_to_random_state=True)
        self._svc = None
        self.X_train = None
        self.y_train = None
        self.y_finite = None
        self.at_least_one_finite = False
        self.all_finite = False
        self.diff_threshold = None
        self._max_y = None
        self._fit_count = 0        # bumped by every fit: what the device-side copy of the decision function is keyed by

    @property
    def d(self):
        if self.X_train is None:
            raise ValueError("You need to add some data before determining its dimension.")
        return self.X_train.shape[1]

    @property
    def n(self):
        return 0 if self.y_train is None else len(self.y_train)

    @property
    def abs_threshold(self):
        return self._max_y - self.diff_threshold

    @staticmethod
    def _is_finite_raw(y, diff_threshold, max_y=None):
        if max_y is None:
            max_y = np.max(y)
        return np.greater_equal(y, max_y - diff_threshold) & np.isfinite(y)

    def is_finite(self, y):
        if self.y_train is None:
            raise ValueError("Cannot do anything: the SVM has not been trained yet!")
        return self._is_finite_raw(y, self.diff_threshold, self._max_y)

    def fit(self, X, y, diff_threshold):
        from sklearn.svm import SVC
        self._fit_count = getattr(self, "_fit_count", 0) + 1
        self.X_train, self.y_train = np.copy(X), np.copy(y)
        if np.all(self.y_train == -np.inf):
            self.at_least_one_finite = False
            self.y_finite = np.full(len(X), False)
            return self.y_finite
        self.at_least_one_finite = True
        self.diff_threshold = diff_threshold
        self._max_y = max(self.y_train)
        self.y_finite = self._is_finite_raw(self.y_train, diff_threshold, max_y=self._max_y)
        self.all_finite = bool(np.all(self.y_finite))
        if not self.all_finite:
            self._svc = SVC(random_state=self.random_state, **self._svc_args)
            self._svc.fit(self.X_train, self.y_finite)
        return self.y_finite

    def device_params(self):
        """``(support_vectors, dual_coef, gamma, intercept, positive_is_finite)`` of the fitted
        two-class RBF SVC, with ``predict(x) == (sum_i coef_i exp(-gamma |x - sv_i|^2) + intercept
        > 0) == positive_is_finite``; ``None`` when there is nothing to evaluate on the device
        (not trained, all points finite or none, other kernel)."""
        svc = self._svc
        if (self.y_train is None or self.all_finite or not self.at_least_one_finite or svc is None
                or svc.kernel != "rbf" or len(svc.classes_) != 2):
            return None
        return (np.ascontiguousarray(svc.support_vectors_, dtype=float),
                np.ascontiguousarray(svc.dual_coef_[0], dtype=float), float(svc._gamma),
                float(svc.intercept_[0]), bool(svc.classes_[1]))

    def predict(self, X, validate=True):
        if self.y_train is None:
            raise ValueError("The SVM has not been trained yet.")
        X = np.atleast_2d(X)
        if self.all_finite:
            return np.full(len(X), True)
        if not self.at_least_one_finite:
            warnings.warn("Only -inf points added to the classifier so far. "
                          "Returning False unconditionally.")
            return np.full(len(X), False)
        return self._svc.predict(X).astype(bool)
