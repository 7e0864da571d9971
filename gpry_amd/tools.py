"""Small host-side helpers with the semantics of ``gpry/tools.py`` that the hot path uses."""
import numpy as np
from scipy.special import erfc
from scipy.stats import chi2


def check_random_state(seed, convert_to_random_state=False):
    """``gpry/tools.py:134-145``: a numpy ``Generator`` passes through (or is wrapped into a ``RandomState`` over the same
    bit generator, for scikit-learn's estimators); ``None``, ints and ``RandomState`` follow scikit-learn's rule."""
    rs_type = np.random.RandomState
    if isinstance(seed, np.random.Generator):
        return rs_type(seed.bit_generator) if convert_to_random_state else seed
    if isinstance(seed, rs_type):
        return seed
    if seed is None or seed is np.random:
        return np.random.mtrand._rand           # numpy's global stream
    if isinstance(seed, (int, np.integer)):
        return rs_type(seed)
    raise ValueError(f"{seed!r} cannot be used to seed a numpy.random.RandomState instance")


def get_random_generator(seed=None):
    """Single-process form of ``gpry/mpi.py:31-50``: the first child of the seed sequence."""
    if not isinstance(seed, np.random.Generator):
        seed = np.random.default_rng(np.random.SeedSequence(seed).spawn(1)[0])
    return seed


def nstd_of_1d_nstd(n1, d):
    """``gpry/tools.py:100-109``."""
    return np.sqrt(chi2.isf(erfc(n1 / np.sqrt(2)), d))


def delta_logp_of_1d_nstd(n1, d):
    """``gpry/tools.py:112-118``."""
    return 0.5 * nstd_of_1d_nstd(n1, d) ** 2


def generic_params_names(n, prefix="x_"):
    return [f"{prefix}{k}" for k in range(1, int(n) + 1)]


def get_Xnumber(value, X_letter, X_value=None, dtype=int, varname=None):
    """Numbers with a symbolic factor: ``"5d"`` = 5 x d, ``"30d1.5"`` = 30 x d^1.5, ``"d"`` = d, plain numbers as they
    are (``gpry/tools.py:185-234``).  With ``X_value=None`` nothing is multiplied and the parts come back as
    ``(number, has_X, power)``."""
    coefficient, has_X, power = value, False, None
    if isinstance(value, str) and X_letter in value:
        has_X = True
        head, _, tail = value.partition(X_letter)
        coefficient, power = head or 1, tail or None
    try:
        coefficient = float(coefficient)
        exponent = None if power is None else float(power)
        if X_value is None:
            return dtype(coefficient), has_X, exponent
        factor = 1 if not has_X else X_value if exponent is None else X_value ** exponent
        return dtype(coefficient * factor)
    except (ValueError, TypeError) as excpt:
        where = f"Error setting variable '{varname}': " if varname else ""
        raise ValueError(f"{where}Could not convert {value} of type {type(value)} into type {dtype.__name__}. Pass "
                         f"either a string ending in '{X_letter}' or a valid {dtype.__name__} value.") from excpt


def check_and_return_bounds(bounds):
    """``bounds`` as a (d, 2) array, or TypeError."""
    try:
        box = np.atleast_2d(bounds)
        ok = box.ndim == 2 and box.shape[1] == 2
    except ValueError:           # ragged input
        ok = False
    if not ok:
        raise TypeError(f"bounds must be a (dim, 2) array of bounds, but is {bounds}")
    return box


def _same_dimension(box, pts, what, exc):
    if box.shape[0] != pts.shape[1]:
        raise exc(f"bounds and {what} appear to have different dimensionalities: "
                  f"{box.shape[0]} for bounds and {pts.shape[1]} for {what}.")


def is_in_bounds(points, bounds, check_shape=False):
    """``gpry/tools.py:263-290``: closed box test per row."""
    pts = np.atleast_2d(points)
    if check_shape:
        bounds = check_and_return_bounds(bounds)
        _same_dimension(bounds, pts, "point", ValueError)
    inside = (pts >= bounds[:, 0]) & (pts <= bounds[:, 1])
    return inside.all(axis=1)


def shrink_bounds(bounds, samples, factor=1):
    """Smallest box around ``samples``, widened about its centre by ``factor`` and clipped to ``bounds``
    (``gpry/tools.py:308-360``)."""
    box, pts = check_and_return_bounds(bounds), np.atleast_2d(samples)
    _same_dimension(box, pts, "samples", TypeError)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    margin = (factor - 1) / 2 * (hi - lo)
    return np.column_stack((np.maximum(lo - margin, box[:, 0]), np.minimum(hi + margin, box[:, 1]))).astype(float)


def remove_0_weight_samples(weights, *arrays):
    """``weights`` and every array (``None`` passes through) without the rows of zero weight (``gpry/tools.py:400-417``)."""
    if any(a is not None and a.shape[0] != len(weights) for a in arrays):
        raise ValueError("weights and some of the arrays have different lengths.")
    zero = np.flatnonzero(weights == 0)
    return [np.delete(weights, zero)] + [None if a is None else np.delete(a, zero, axis=0) for a in arrays]
