"""Affine X / y maps used by the GP (host side; the device applies the same maps fused
into its kernels through ``gpry_affine``).  Interfaces follow ``gpry/preprocessing.py``:
``Normalize_bounds`` (:311-411), ``Normalize_y`` (:528-630), ``DummyPreprocessor`` (:29-55).

All three are ``v -> (v - origin) / width`` with a different source of (origin, width): nothing, the prior box, the
moments of the finite training targets.  One base class carries the four directions of the map; the arithmetic
(subtract then divide, multiply then add) is the reference's, so the transformed values have its bits.
"""
import numpy as np


def _unchanged(v, *unused, **unused_kw):
    return v


class DummyPreprocessor:
    """The identity.  GPry passes the class itself around as well as instances, so nothing here binds ``self``."""
    is_linear = True
    fitted = True
    fit = staticmethod(lambda *args, **kwargs: None)
    transform = inverse_transform = staticmethod(_unchanged)
    transform_scale = inverse_transform_scale = staticmethod(_unchanged)
    transform_bounds = staticmethod(_unchanged)


class _AffineMap:
    """``origin_width()`` -> (origin, width), per dimension or scalar."""
    is_linear = True

    def origin_width(self):
        raise NotImplementedError

    def transform(self, v):
        origin, width = self.origin_width()
        return (v - origin) / width

    def inverse_transform(self, v):
        origin, width = self.origin_width()
        return (v * width) + origin

    def transform_scale(self, v):
        return v / self.origin_width()[1]

    def inverse_transform_scale(self, v):
        return v * self.origin_width()[1]


class Normalize_bounds(_AffineMap):
    """Prior box -> unit cube."""
    fitted = True

    def __init__(self, bounds):
        self.update_bounds(bounds)

    def update_bounds(self, bounds):
        box = np.asarray(bounds)
        lo, hi = box[:, 0], box[:, 1]
        if (lo > hi).any():
            raise ValueError(f"The bounds must be in dimension-wise order min->max, got \n{box}")
        self.bounds, self.bounds_min, self.bounds_max = box, lo, hi

    def origin_width(self):
        return self.bounds_min, self.bounds_max - self.bounds_min

    def transform_bounds(self, bounds):
        unit = np.ones_like(bounds)
        unit[:, 0] = 0
        return unit

    def fit(self, X, y):
        """The map is fixed by the prior box."""


class Normalize_y(_AffineMap):
    """Centre and scale of the finite training targets: mean / population std, or median / inter-quartile range."""

    def __init__(self, use_median=False):
        self.use_median = bool(use_median)
        self.mean_ = self.std_ = None

    @property
    def fitted(self):
        return not (self.mean_ is None or self.std_ is None)

    def fit(self, X, y):
        finite = y[np.isfinite(y)]
        if self.use_median:
            quartiles = np.percentile(finite, [25, 50, 75])
            self.mean_, self.std_ = quartiles[1], quartiles[2] - quartiles[0]
        else:
            self.mean_, self.std_ = np.mean(finite), np.std(finite)

    def origin_width(self):
        if not self.fitted:
            raise TypeError("mean_ and std_ have not been fit before")
        return self.mean_, self.std_
