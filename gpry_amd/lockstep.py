"""Several L-BFGS-B runs advanced side by side, their objective evaluations batched.

GPry's acquisition optimiser (gpry/gp_acquisition.py:270-389, 391-497) starts ``n_restarts_optimizer`` runs of scipy's
``fmin_l_bfgs_b`` one after another; every step of every run is one posterior evaluation with x-gradients of ONE point.
The runs are independent -- the optimiser draws no random numbers -- so they can be stepped together: each round collects
the points the runs want evaluated and hands them to ONE batched call (``gpry_predict_grad_batch`` on the device reads
V = L^-1 once for all of them instead of twice per point).

The runs ARE scipy's: this module drives the same compiled routine (``scipy.optimize._lbfgsb.setulb``, reverse
communication) with the same workspace set-up, tolerances and stopping rules as ``scipy.optimize._lbfgsb_py._minimize_lbfgsb``
of the scipy this was written against (1.15); given the same function values a run takes the same steps as
``fmin_l_bfgs_b``.  That entry point is private, so ``available()`` decides once per process whether it may be used:

1. by introspection, not by version number alone: the routine's argument list (its docstring) must be the one called
   below, and the work arrays that scipy's OWN wrapper allocates for it (read from the source of
   ``_lbfgsb_py._minimize_lbfgsb``) must have the sizes used here -- a routine that expects larger work arrays would write
   past ours.  A scipy inside ``SCIPY_TESTED`` whose source cannot be read is accepted on its version; any other scipy is
   accepted only if both checks pass;
2. by a dry run with canaries: the self-check drives two small bounded problems through work arrays that are views of
   larger buffers filled with a canary value -- the padding must come back untouched -- and the runs must reproduce
   ``fmin_l_bfgs_b`` bit for bit.

Callers fall back to the one-after-another form otherwise and SAY SO: ``why()`` gives the reason, ``warn_once()`` emits it
as one ``UserWarning`` per process, and the fit records it in ``fit_stats`` (``gpry_amd/gpr.py``).
"""
import re
import warnings

import numpy as np

_STATE = {"checked": False, "ok": False, "why": "", "warned": False, "how": ""}

# scipy releases this driver was run against (the C translation that replaced the Fortran routine in 1.15: ``wa`` of
# 2mn + 5n + 11m^2 + 8m doubles, ``iwa`` of 3n, ``isave`` of 44 and ``dsave`` of 29 entries, two-word ``task``)
SCIPY_TESTED = ((1, 15), (1, 16))       # [first, one past the last) minor version

_SETULB_ARGS = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)"
# allocations in scipy's own wrapper that must match the ones in _Run (whitespace-insensitive)
_WRAPPER_ALLOCATIONS = (
    r"wa=zeros\(2\*m\*n\+5\*n\+11\*m\*m\+8\*m,float64\)",
    r"iwa=zeros\(3\*n,dtype=np\.int32\)",
    r"task=zeros\(2,dtype=np\.int32\)",
    r"ln_task=zeros\(2,dtype=np\.int32\)",
    r"lsave=zeros\(4,dtype=np\.int32\)",
    r"isave=zeros\(44,dtype=np\.int32\)",
    r"dsave=zeros\(29,dtype=float64\)",
    r"_lbfgsb\.setulb\(m,x,low_bnd,upper_bnd,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task\)",
)
_CANARY_F, _CANARY_I, _PAD = -7.25e300, -0x5A5A5A5, 4096


def _scipy_version_ok():
    import scipy
    try:
        ver = tuple(int(p) for p in scipy.__version__.split(".")[:2])
    except ValueError:
        return False, scipy.__version__
    return SCIPY_TESTED[0] <= ver < SCIPY_TESTED[1], scipy.__version__


def _introspect():
    """(ok, detail): does the installed routine take the arguments and work arrays this driver passes?"""
    try:
        from scipy.optimize import _lbfgsb, _lbfgsb_py
    except Exception as e:
        return False, f"scipy.optimize._lbfgsb cannot be imported ({e!r})"
    if not hasattr(_lbfgsb, "setulb"):
        return False, "scipy.optimize._lbfgsb has no setulb"
    doc = (getattr(_lbfgsb.setulb, "__doc__", None) or "").replace(" ", "")
    if doc and _SETULB_ARGS not in doc:
        return False, f"setulb has another argument list: {doc.splitlines()[0][:160]!r}"
    try:
        import inspect
        src = re.sub(r"\s+", "", inspect.getsource(_lbfgsb_py._minimize_lbfgsb))
    except Exception as e:
        return None, f"the source of scipy's wrapper cannot be read ({e!r})"
    missing = [p for p in _WRAPPER_ALLOCATIONS if not re.search(p, src)]
    if missing:
        return False, ("scipy's own wrapper sets the routine up differently (not found: "
                       + ", ".join(m.replace("\\", "") for m in missing[:3]) + ")")
    return True, "argument list and work-array sizes match scipy's own wrapper"


def _padded(n, dtype, pad):
    """zeros(n) as a view of a buffer with `pad` canary entries behind it (pad = 0: a plain array)"""
    if not pad:
        return np.zeros(n, dtype=dtype), None
    buf = np.full(n + pad, _CANARY_F if dtype == np.float64 else _CANARY_I, dtype=dtype)
    buf[:n] = 0
    return buf[:n], buf


class _Run:
    __slots__ = ("x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave", "n_iter", "nfev", "done",
                 "seen_x", "seen_f", "seen_g", "_bufs")

    def __init__(self, x0, n, m, pad=0):
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.float64)
        self.g = np.zeros(n, dtype=np.float64)
        self._bufs = []
        for name, size, dt in (("wa", 2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64), ("iwa", 3 * n, np.int32),
                               ("task", 2, np.int32), ("ln_task", 2, np.int32), ("lsave", 4, np.int32),
                               ("isave", 44, np.int32), ("dsave", 29, np.float64)):
            view, buf = _padded(size, dt, pad)
            setattr(self, name, view)
            if buf is not None:
                self._bufs.append((name, size, buf))
        self.n_iter = 0
        self.nfev = 0
        self.done = False
        self.seen_x = self.seen_f = self.seen_g = None      # the last point evaluated (scipy's ScalarFunction keeps it too)

    def canaries_intact(self):
        for name, size, buf in self._bufs:
            canary = _CANARY_F if buf.dtype == np.float64 else _CANARY_I
            if not (buf[size:] == canary).all():
                return False, name
        return True, ""


def minimize_lockstep(fg_batch, X0, bounds, m=10, factr=1e7, pgtol=1e-5, maxfun=15000, maxiter=15000, maxls=20, _pad=0,
                      _runs_out=None):
    """Minimise ``k`` starts at once.  ``fg_batch(X)`` -> ``(f (j,), g (j, n))`` for the ``j <= k`` rows of ``X``;
    ``bounds``: ``(n, 2)`` with +-inf for open sides.  Returns ``(X (k, n), F (k,), nfev (k,))``.
    (``_pad`` / ``_runs_out``: the self-check's canary padding behind every work array, and the run objects it inspects.)"""
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
    runs = [_Run(np.clip(x0, lo, hi), n, m, _pad) for x0 in X0]
    if _runs_out is not None:
        _runs_out.extend(runs)
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
    """True if scipy's reverse-communication routine is there, takes what this driver passes (introspection) and the
    driver reproduces ``fmin_l_bfgs_b`` with it (dry run behind canary padding).  Decided once per process; ``why()``
    says why not."""
    if _STATE["checked"]:
        return _STATE["ok"]
    _STATE["checked"] = True
    in_range, ver = _scipy_version_ok()
    looks_ok, detail = _introspect()
    if looks_ok is False:
        _STATE["why"] = f"scipy {ver}: {detail}"
        return False
    if looks_ok is None and not in_range:       # nothing to go by: neither a tested release nor a readable wrapper
        _STATE["why"] = (f"scipy {ver} is outside the releases this driver of its private L-BFGS-B routine was tested with "
                         f"({SCIPY_TESTED[0][0]}.{SCIPY_TESTED[0][1]} <= version < {SCIPY_TESTED[1][0]}.{SCIPY_TESTED[1][1]}) "
                         f"and {detail}")
        return False
    try:
        import scipy.optimize
        A = np.array([[3.0, 0.4, 0.1], [0.4, 2.0, -0.3], [0.1, -0.3, 1.5]])
        c = np.array([1.0, -2.0, 0.5])

        def fg(x):
            return 0.5 * x @ A @ x - c @ x + 0.1 * np.sum(x ** 4), A @ x - c + 0.4 * x ** 3

        bnds = np.array([[-0.2, 0.3], [-1.0, 1.0], [-np.inf, np.inf]])
        starts = np.array([[0.25, -0.9, 2.0], [-0.1, 0.5, -1.0]])
        runs = []
        got_x, got_f, _ = minimize_lockstep(lambda X: (np.array([fg(x)[0] for x in X]), np.array([fg(x)[1] for x in X])),
                                            starts, bnds, _pad=_PAD, _runs_out=runs)
        for r in runs:
            intact, name = r.canaries_intact()
            if not intact:
                raise RuntimeError(f"the routine wrote past the end of `{name}`: its work arrays are larger than scipy 1.15's")
        for s, gx, gf in zip(starts, got_x, got_f):
            rx, rf, _ = scipy.optimize.fmin_l_bfgs_b(fg, s, bounds=[tuple(b) for b in bnds], approx_grad=False)
            if not (np.array_equal(rx, gx) and rf == gf):
                raise RuntimeError("the lock-step driver does not reproduce fmin_l_bfgs_b")
        _STATE["ok"] = True
        _STATE["how"] = f"scipy {ver}: {detail}; dry run behind canary padding reproduces fmin_l_bfgs_b"
    except Exception as e:          # private API moved, other scipy: the callers run the restarts one after another
        _STATE["why"] = f"scipy {ver}: {e!r}"
        _STATE["ok"] = False
    return _STATE["ok"]


def why():
    """Why ``available()`` is False ('' if it is True or has not been asked)."""
    return "" if _STATE["ok"] else _STATE["why"]


def how():
    """What ``available()`` accepted the routine on ('' if it did not)."""
    return _STATE["how"] if _STATE["ok"] else ""


def warn_once(what="multi-restart fits"):
    """One UserWarning per process when the side-by-side form is not available: the caller is about to run 3-7x slower."""
    if _STATE["checked"] and not _STATE["ok"] and not _STATE["warned"]:
        _STATE["warned"] = True
        warnings.warn(f"gpry_amd: {what} run one restart after another instead of side by side (3-7x slower): "
                      f"{_STATE['why']}.  Results are unchanged.", UserWarning, stacklevel=3)
