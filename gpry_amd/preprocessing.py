"""Affine X / y maps used by the GP (host side; the device applies the same maps fused
into its kernels through ``gpry_affine``).  Interfaces follow ``gpry/preprocessing.py``:
``Normalize_bounds`` (:311-411), ``Normalize_y`` (:528-630), ``DummyPreprocessor`` (:29-55).
"""
import numpy as np


class DummyPreprocessor:
    """Identity map; usable as a class (as the reference does) or as an instance."""
    is_linear = True
    fitted = True

    @classmethod
    def fit(cls, *args, **kwargs):
        return None

    @classmethod
    def transform_bounds(cls, bounds):
        return bounds

    @classmethod
    def transform(cls, v):
        return v

    @classmethod
    def inverse_transform(cls, v):
        return v

    @classmethod
    def transform_scale(cls, v):
        return v

    @classmethod
    def inverse_transform_scale(cls, v):
        return v


class Normalize_bounds:
    """Maps the prior box onto the unit cube: ``x_ = (x - lo) / (hi - lo)``."""
    is_linear = True

    def __init__(self, bounds):
        self.update_bounds(bounds)
        self.fitted = True

    def update_bounds(self, bounds):
        bounds = np.asarray(bounds)
        if np.any(bounds[:, 0] > bounds[:, 1]):
            raise ValueError(f"The bounds must be in dimension-wise order min->max, got \n{bounds}")
        self.bounds = bounds
        self.bounds_min = bounds[:, 0]
        self.bounds_max = bounds[:, 1]

    def transform_bounds(self, bounds):
        out = np.ones_like(bounds)
        out[:, 0] = 0
        return out

    def fit(self, X, y):
        """Nothing to fit: the map is fixed by the prior bounds."""

    def transform(self, X):
        return (X - self.bounds_min) / (self.bounds_max - self.bounds_min)

    def inverse_transform(self, X):
        return (X * (self.bounds_max - self.bounds_min)) + self.bounds_min

    def inverse_transform_scale(self, X):
        return X * (self.bounds_max - self.bounds_min)


class Normalize_y:
    """Standardises targets with the mean / population std of the finite training values
    (or median / inter-quartile range with ``use_median``)."""
    is_linear = True

    def __init__(self, use_median=False):
        self.mean_ = None
        self.std_ = None
        self.use_median = bool(use_median)

    @property
    def fitted(self):
        return self.mean_ is not None and self.std_ is not None

    def _require_fit(self):
        if not self.fitted:
            raise TypeError("mean_ and std_ have not been fit before")

    def fit(self, X, y):
        y = y[np.isfinite(y)]
        if self.use_median:
            q25, q50, q75 = np.percentile(y, [25, 50, 75])
            self.mean_, self.std_ = q50, q75 - q25
        else:
            self.mean_, self.std_ = np.mean(y), np.std(y)

    def transform(self, y):
        self._require_fit()
        return (y - self.mean_) / self.std_

    def inverse_transform(self, y):
        self._require_fit()
        return (y * self.std_) + self.mean_

    def transform_scale(self, scale):
        self._require_fit()
        return scale / self.std_

    def inverse_transform_scale(self, scale):
        self._require_fit()
        return scale * self.std_
