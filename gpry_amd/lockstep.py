"""Several L-BFGS-B runs advanced side by side, their objective evaluations batched.

GPry's acquisition optimiser (gpry/gp_acquisition.py:270-389, 391-497) starts ``n_restarts_optimizer`` runs of scipy's
``fmin_l_bfgs_b`` one after another; every step of every run is one posterior evaluation with x-gradients of ONE point.
The runs are independent -- the optimiser draws no random numbers -- so they can be stepped together: each round collects
the points the runs want evaluated and hands them to ONE batched call (``gpry_predict_grad_batch`` on the device reads
V = L^-1 once for all of them instead of twice per point).

The runs ARE scipy's: this module drives the same compiled routine (``scipy.optimize._lbfgsb.setulb``, reverse
communication) with the same workspace set-up, tolerances and stopping rules as ``scipy.optimize._lbfgsb_py._minimize_lbfgsb``
of the scipy this was written against (1.15); given the same function values a run takes the same steps as
``fmin_l_bfgs_b``.  That entry point is private: ``available()`` first refuses every scipy outside ``SCIPY_TESTED`` (the
work-array sizes are hard-coded: a routine that wants larger ones must never be called), then checks once per process
that a run through this driver reproduces ``fmin_l_bfgs_b`` bit for bit on a small bounded problem; callers fall back to
the one-after-another form otherwise.
"""
import numpy as np

_STATE = {"checked": False, "ok": False, "why": ""}

# scipy releases whose ``_lbfgsb.setulb`` has the argument list and workspace sizes used below (the C translation that
# replaced the Fortran routine in 1.15: ``wa`` of 2mn + 5n + 11m^2 + 8m doubles, ``iwa`` of 3n, ``isave`` of 44 and
# ``dsave`` of 29 entries, two-word ``task``).  A routine that expects larger work arrays would write past ours before
# any self-check could notice, so other versions are refused BEFORE the first call.
SCIPY_TESTED = ((1, 15), (1, 16))       # [first, one past the last) minor version


def _scipy_version_ok():
    import scipy
    try:
        ver = tuple(int(p) for p in scipy.__version__.split(".")[:2])
    except ValueError:
        return False, scipy.__version__
    return SCIPY_TESTED[0] <= ver < SCIPY_TESTED[1], scipy.__version__


class _Run:
    __slots__ = ("x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave", "n_iter", "nfev", "done",
                 "seen_x", "seen_f", "seen_g")

    def __init__(self, x0, n, m):
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.float64)
        self.g = np.zeros(n, dtype=np.float64)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.n_iter = 0
        self.nfev = 0
        self.done = False
        self.seen_x = self.seen_f = self.seen_g = None      # the last point evaluated (scipy's ScalarFunction keeps it too)


def minimize_lockstep(fg_batch, X0, bounds, m=10, factr=1e7, pgtol=1e-5, maxfun=15000, maxiter=15000, maxls=20):
    """Minimise ``k`` starts at once.  ``fg_batch(X)`` -> ``(f (j,), g (j, n))`` for the ``j <= k`` rows of ``X``;
    ``bounds``: ``(n, 2)`` with +-inf for open sides.  Returns ``(X (k, n), F (k,), nfev (k,))``."""
    from scipy.optimize import _lbfgsb
    X0 = np.atleast_2d(np.asarray(X0, dtype=float))
    k, n = X0.shape
    bounds = np.asarray(bounds, dtype=float)
    if bounds.shape != (n, 2):
        raise ValueError("length of x0 != length of bounds")
    if (bounds[:, 0] > bounds[:, 1]).any():
        raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
    lo, hi = bounds[:, 0], bounds[:, 1]
    nbd = np.zeros(n, np.int32)
    low_bnd, upper_bnd = np.zeros(n), np.zeros(n)
    code = {(False, False): 0, (True, False): 1, (True, True): 2, (False, True): 3}
    for i in range(n):
        has_lo, has_hi = not np.isinf(lo[i]), not np.isinf(hi[i])
        if has_lo:
            low_bnd[i] = lo[i]
        if has_hi:
            upper_bnd[i] = hi[i]
        nbd[i] = code[has_lo, has_hi]
    runs = [_Run(np.clip(x0, lo, hi), n, m) for x0 in X0]
    active = list(range(k))
    while active:
        waiting = []
        for i in active:
            r = runs[i]
            while True:
                _lbfgsb.setulb(m, r.x, low_bnd, upper_bnd, nbd, r.f, r.g, factr, pgtol, r.wa, r.iwa, r.task, r.lsave,
                               r.isave, r.dsave, maxls, r.ln_task)
                if r.task[0] == 3:                     # wants f and g at r.x
                    if r.seen_x is not None and not (r.x != r.seen_x).any():
                        # the routine asks again for the point it was just given (scipy's wrapper answers from its cache
                        # without calling the function: neither do we)
                        r.f[()] = r.seen_f
                        r.g[:] = r.seen_g
                        continue
                    waiting.append(i)
                    break
                if r.task[0] == 1:                     # a new iteration has started
                    r.n_iter += 1
                    if r.n_iter >= maxiter:
                        r.task[0], r.task[1] = 5, 504
                    elif r.nfev > maxfun:
                        r.task[0], r.task[1] = 5, 502
                    continue
                r.done = True
                break
        if waiting:
            F, G = fg_batch(np.array([runs[i].x for i in waiting]))
            for j, i in enumerate(waiting):
                r = runs[i]
                r.f[()] = F[j]
                r.g[:] = G[j]
                if r.seen_x is None:
                    r.seen_x, r.seen_g = r.x.copy(), r.g.copy()
                else:
                    r.seen_x[:] = r.x
                    r.seen_g[:] = r.g
                r.seen_f = float(F[j])
                r.nfev += 1
        active = [i for i in active if not runs[i].done]
    return (np.array([r.x for r in runs]), np.array([float(r.f) for r in runs]),
            np.array([r.nfev for r in runs]))


def available():
    """True if scipy's reverse-communication routine is there and this driver reproduces ``fmin_l_bfgs_b`` with it."""
    if _STATE["checked"]:
        return _STATE["ok"]
    _STATE["checked"] = True
    ok, ver = _scipy_version_ok()
    if not ok:
        _STATE["why"] = (f"scipy {ver} is outside the range this driver of its private L-BFGS-B routine was tested with "
                         f"({SCIPY_TESTED[0][0]}.{SCIPY_TESTED[0][1]} <= version < {SCIPY_TESTED[1][0]}.{SCIPY_TESTED[1][1]})")
        return False
    try:
        import scipy.optimize
        A = np.array([[3.0, 0.4, 0.1], [0.4, 2.0, -0.3], [0.1, -0.3, 1.5]])
        c = np.array([1.0, -2.0, 0.5])

        def fg(x):
            return 0.5 * x @ A @ x - c @ x + 0.1 * np.sum(x ** 4), A @ x - c + 0.4 * x ** 3

        bnds = np.array([[-0.2, 0.3], [-1.0, 1.0], [-np.inf, np.inf]])
        starts = np.array([[0.25, -0.9, 2.0], [-0.1, 0.5, -1.0]])
        got_x, got_f, _ = minimize_lockstep(lambda X: (np.array([fg(x)[0] for x in X]), np.array([fg(x)[1] for x in X])),
                                            starts, bnds)
        for s, gx, gf in zip(starts, got_x, got_f):
            rx, rf, _ = scipy.optimize.fmin_l_bfgs_b(fg, s, bounds=[tuple(b) for b in bnds], approx_grad=False)
            if not (np.array_equal(rx, gx) and rf == gf):
                raise RuntimeError("the lock-step driver does not reproduce fmin_l_bfgs_b")
        _STATE["ok"] = True
    except Exception as e:          # private API moved, other scipy: the callers run the restarts one after another
        _STATE["why"] = repr(e)
        _STATE["ok"] = False
    return _STATE["ok"]
