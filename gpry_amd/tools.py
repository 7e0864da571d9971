"""Small host-side helpers with the semantics of ``gpry/tools.py`` that the hot path uses."""
import numpy as np
from scipy.special import erfc
from scipy.stats import chi2


def check_random_state(seed, convert_to_random_state=False):
    """``gpry/tools.py:134-145``: numpy Generators pass through, the rest goes to sklearn's rule."""
    if isinstance(seed, np.random.Generator):
        if convert_to_random_state:
            return np.random.RandomState(seed.bit_generator)
        return seed
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError(f"{seed!r} cannot be used to seed a numpy.random.RandomState instance")


def get_random_generator(seed=None):
    """Single-process form of ``gpry/mpi.py:31-50`` (first spawned child seed)."""
    if isinstance(seed, np.random.Generator):
        return seed
    return np.random.default_rng(np.random.SeedSequence(seed).spawn(1)[0])


def nstd_of_1d_nstd(n1, d):
    """``gpry/tools.py:100-109``."""
    return np.sqrt(chi2.isf(erfc(n1 / np.sqrt(2)), d))


def delta_logp_of_1d_nstd(n1, d):
    """``gpry/tools.py:112-118``."""
    return 0.5 * nstd_of_1d_nstd(n1, d) ** 2


def generic_params_names(n, prefix="x_"):
    return [prefix + str(i + 1) for i in range(int(n))]


def get_Xnumber(value, X_letter, X_value=None, dtype=int, varname=None):
    """Parse numbers like ``"5d"`` or ``"30d1.5"`` (``gpry/tools.py:185-234``).

    With ``X_value=None`` returns ``(number, has_X, power)``.
    """
    if value == X_letter:
        value = "1" + X_letter
    has_X, power, num = False, None, value
    if isinstance(value, str) and X_letter in value:
        has_X = True
        num, power = value.split(X_letter)
        num = num or 1
        power = power or None
    try:
        num = float(num)
        if X_value is None:
            return dtype(num), has_X, (None if power is None else float(power))
        mult = 1
        if has_X:
            mult = X_value if power is None else X_value ** float(power)
        return dtype(num * mult)
    except (ValueError, TypeError) as excpt:
        pre = f"Error setting variable '{varname}': " if varname else ""
        raise ValueError(
            pre + f"Could not convert {value} of type {type(value)} into type {dtype.__name__}. "
            f"Pass either a string ending in '{X_letter}' or a valid {dtype.__name__} value."
        ) from excpt


def check_and_return_bounds(bounds):
    try:
        b = np.atleast_2d(bounds)
        if b.shape[1] != 2:
            raise ValueError
    except ValueError as excpt:
        raise TypeError(f"bounds must be a (dim, 2) array of bounds, but is {bounds}") from excpt
    return b


def is_in_bounds(points, bounds, check_shape=False):
    """``gpry/tools.py:263-290``: closed box test per row."""
    points = np.atleast_2d(points)
    if check_shape:
        bounds = check_and_return_bounds(bounds)
        if bounds.shape[0] != points.shape[1]:
            raise ValueError("bounds and point appear to have different dimensionalities: "
                             f"{bounds.shape[0]} for bounds and {points.shape[1]} for point.")
    return np.all((points >= bounds[:, 0]) & (points <= bounds[:, 1]), axis=1)


def shrink_bounds(bounds, samples, factor=1):
    """Smallest box around ``samples`` scaled by ``factor``, clipped to ``bounds``
    (``gpry/tools.py:308-360``)."""
    bounds = check_and_return_bounds(bounds)
    samples = np.atleast_2d(samples)
    if bounds.shape[0] != samples.shape[1]:
        raise TypeError("bounds and samples appear to have different dimensionalities: "
                        f"{bounds.shape[0]} for bounds and {samples.shape[1]} for samples.")
    lo, hi = samples.min(axis=0), samples.max(axis=0)
    delta = (factor - 1) / 2 * (hi - lo)
    out = np.empty(bounds.shape, dtype=float)
    out[:, 0] = np.maximum(lo - delta, bounds[:, 0])
    out[:, 1] = np.minimum(hi + delta, bounds[:, 1])
    return out


def remove_0_weight_samples(weights, *arrays):
    """``gpry/tools.py:400-417``."""
    drop = np.where(weights == 0)[0]
    out = [np.delete(weights, drop)]
    for a in arrays:
        if a is None:
            out.append(None)
        elif a.shape[0] != len(weights):
            raise ValueError("weights and some of the arrays have different lengths.")
        else:
            out.append(np.delete(a, drop, axis=0))
    return out
