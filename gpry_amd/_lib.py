"""ctypes binding of ``libgpry_hip.so`` (C ABI declared in ``include/gpry_hip.h``).

There is deliberately no CPU fallback: if the shared library is missing, or a device
call is made on a machine without a GPU, the error is raised to the caller.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpry_hip.so")

GPRY_MAX_DIM = 32      # include/gpry_hip.h: ABI array size and the enforced limit on d
MASK_CLASSIFIED_INF = 1
MASK_OUTSIDE_TRUST = 2

KERNEL_IDS = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3}


class GpryAffine(C.Structure):
    _fields_ = [("has_x_affine", C.c_int),
                ("x_lo", C.c_double * GPRY_MAX_DIM),
                ("x_span", C.c_double * GPRY_MAX_DIM),
                ("y_mean", C.c_double), ("y_std", C.c_double),
                ("clip_hi", C.c_double)]


class GpryCand(C.Structure):
    _fields_ = [("acq", C.c_double), ("y", C.c_double), ("sigma", C.c_double),
                ("idx", C.c_int64)]


CAND_DTYPE = np.dtype([("acq", "<f8"), ("y", "<f8"), ("sigma", "<f8"), ("idx", "<i8")])

_P = C.POINTER
_dp = _P(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/gpry_hip.h one to one
SIGNATURES = {
    "gpry_version": (C.c_int, []),
    "gpry_device_count": (C.c_int, [_P(C.c_int)]),
    "gpry_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, _P(C.c_int64), _P(C.c_int),
                                   _P(C.c_int), C.c_char_p, C.c_int]),
    "gpry_ctx_create": (C.c_int, [C.c_int, _P(_vp)]),
    "gpry_ctx_destroy": (C.c_int, [_vp]),
    "gpry_last_error": (C.c_char_p, [_vp]),
    "gpry_ctx_sync": (C.c_int, [_vp]),
    "gpry_ctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "gpry_ctx_get_option": (C.c_int, [_vp, C.c_char_p, _vp]),
    "gpry_set_train": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int]),
    "gpry_set_theta": (C.c_int, [_vp, C.c_int, _vp]),
    "gpry_set_affine": (C.c_int, [_vp, _P(GpryAffine)]),
    "gpry_kernel_train": (C.c_int, [_vp, C.c_int, _vp]),
    "gpry_kernel_cross": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "gpry_factorize": (C.c_int, [_vp, _P(C.c_int)]),
    "gpry_get_factor": (C.c_int, [_vp, _vp, _vp, _vp]),
    "gpry_append_rows": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _P(C.c_int)]),
    "gpry_lml": (C.c_int, [_vp, _vp, C.c_int, _P(C.c_double), _vp, _P(C.c_int)]),
    "gpry_lml_batch": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp]),
    "gpry_predict": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "gpry_debug_serve_stats": (C.c_int, [_vp, _P(C.c_int64), _P(C.c_int64)]),
    "gpry_predict_grad": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "gpry_predict_grad_batch": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp]),
    "gpry_predict_point": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "gpry_set_gates": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, C.c_int, _vp]),
    "gpry_sweep_logexp": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_double, C.c_double,
                                    C.c_double, _vp, _vp, _vp, _P(C.c_int64)]),
    "gpry_sweep_fetch": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "gpry_sweep_topk": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, _P(C.c_int64),
                                  _P(C.c_double)]),
    "gpry_kb_reset": (C.c_int, [_vp]),
    "gpry_kb_register": (C.c_int, [_vp, _vp, C.c_int64, _P(C.c_int64), _vp]),
    "gpry_kb_gram": (C.c_int, [_vp, C.c_int64, _vp, _vp, _P(C.c_int64)]),
    "gpry_comm_unique_id": (C.c_int, [_vp]),
    "gpry_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _P(_vp)]),
    "gpry_comm_destroy": (C.c_int, [_vp]),
    "gpry_comm_info": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int), _P(C.c_int)]),
    "gpry_comm_allgather": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "gpry_comm_allreduce_max": (C.c_int, [_vp, _vp, C.c_int64]),
    "gpry_comm_barrier": (C.c_int, [_vp]),
    "gpry_group_create": (C.c_int, [C.c_int, _P(C.c_int), _vp, _P(_vp)]),
    "gpry_group_destroy": (C.c_int, [_vp]),
    "gpry_group_size": (C.c_int, [_vp, _P(C.c_int), _P(C.c_int)]),
    "gpry_group_member": (_vp, [_vp, C.c_int]),
    "gpry_group_last_error": (C.c_char_p, [_vp]),
    "gpry_group_set_model": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp,
                                       _P(GpryAffine), _P(C.c_int)]),
    "gpry_group_set_gates": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, C.c_int, _vp]),
    "gpry_group_sweep_logexp": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_double, C.c_double,
                                          C.c_double, _vp, _vp, _vp, _P(C.c_int64)]),
    "gpry_group_sweep_fetch": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "gpry_group_sweep_topk": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, _P(C.c_int64),
                                        _P(C.c_double), _P(C.c_int)]),
    "gpry_group_lml_batch": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "gpry_timing_reset": (C.c_int, [_vp]),
    "gpry_timing_get": (C.c_int, [_vp, C.c_char_p, _P(C.c_double), _P(C.c_int64)]),
    "gpry_sweep_info": (C.c_int, [_vp, _P(C.c_int), _vp]),
    "gpry_microbench": (C.c_int, [_vp, C.c_int, C.c_int64, _P(C.c_double)]),
    "gpry_debug_gemm": (C.c_int, [_vp, _vp, _vp, _vp] + [C.c_int] * 9),
    "gpry_debug_logexp": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_double, C.c_double, C.c_double, _vp]),
}


class GpryHipError(RuntimeError):
    """An API / HIP / RCCL failure reported by libgpry_hip.so."""


_lib = None


def load_library(path=None):
    """Load (once) and return the shared library.  Raises if it cannot be loaded."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("GPRY_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise GpryHipError(
            f"{p} not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C gpry_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def device_count():
    n = C.c_int(0)
    lib = load_library()
    if lib.gpry_device_count(C.byref(n)) != 0:
        return 0
    return n.value


def make_affine(x_lo=None, x_span=None, y_mean=0.0, y_std=1.0, clip_hi=np.inf):
    tf = GpryAffine()
    tf.has_x_affine = 0 if x_lo is None else 1
    if x_lo is not None:
        for k in range(len(x_lo)):
            tf.x_lo[k] = float(x_lo[k])
            tf.x_span[k] = float(x_span[k])
    tf.y_mean, tf.y_std = float(y_mean), float(y_std)
    tf.clip_hi = float(clip_hi)
    return tf


class Device:
    """One ``gpry_ctx``: the device-resident GP state of one GPU.  Like the context it wraps, a ``Device`` is not
    re-entrant: one thread at a time (distinct ``Device`` objects may be driven from distinct threads)."""

    applies_gates_in_predict = True     # option "predict_gates": gpry_predict ORs the gates of set_gates into the mask

    def __init__(self, device=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.gpry_ctx_create(int(device), C.byref(self._h))
        if rc != 0:
            msg = self._lib.gpry_last_error(None).decode(errors="replace")
            self._h = None
            raise GpryHipError(f"gpry_ctx_create(device={device}) failed ({rc}): {msg}. "
                               "The HIP path needs a GPU; there is no CPU fallback.")
        self.device = int(device)
        self.N = 0
        self.d = 0

    @classmethod
    def _view(cls, handle, device):
        """Non-owning wrapper of a context that belongs to a ``DeviceGroup``."""
        self = cls.__new__(cls)
        self._lib = load_library()
        self._h = C.c_void_p(handle)
        self._borrowed = True
        self.device, self.N, self.d = int(device), 0, 0
        return self

    # -- plumbing -------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.gpry_last_error(self._h).decode(errors="replace")
            raise GpryHipError(f"{what} failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = None
            return
        if getattr(self, "_h", None):
            try:
                self._lib.gpry_ctx_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def __del__(self):
        self.close()

    def sync(self):
        self._check(self._lib.gpry_ctx_sync(self._h), "gpry_ctx_sync")

    def set_option(self, key, value):
        self._check(self._lib.gpry_ctx_set_option(self._h, key.encode(), int(value)),
                    "gpry_ctx_set_option")

    def get_option(self, key):
        v = C.c_int64(0)
        self._check(self._lib.gpry_ctx_get_option(self._h, key.encode(), C.byref(v)), "gpry_ctx_get_option")
        return v.value

    @property
    def lml_batch_max(self):
        """Largest training set whose ``lml_batch`` runs as ONE chain of launches (option ``lml_batch``; up to 128 points
        the single-launch objective serves a batch whatever this says)."""
        return self.get_option("lml_batch")

    def info(self):
        name = C.create_string_buffer(256)
        arch = C.create_string_buffer(256)
        hbm, ncu, clk = C.c_int64(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.gpry_device_info(self.device, name, 256, C.byref(hbm), C.byref(ncu),
                                               C.byref(clk), arch, 256), "gpry_device_info")
        return {"name": name.value.decode(), "arch": arch.value.decode(),
                "hbm_bytes": hbm.value, "n_cu": ncu.value, "clock_khz": clk.value}

    # -- model state ----------------------------------------------------------------
    def set_train(self, X_, y_, alpha):
        X_ = _f64(X_)
        N, d = X_.shape
        y_ = _f64(y_, (N,))
        alpha = _f64(np.broadcast_to(alpha, (N,)))
        self._check(self._lib.gpry_set_train(self._h, _ptr(X_), _ptr(y_), _ptr(alpha), N, d),
                    "gpry_set_train")
        self.N, self.d = N, d

    def set_theta(self, kernel_id, theta):
        theta = _f64(theta, (self.d + 1,))
        self._check(self._lib.gpry_set_theta(self._h, int(kernel_id), _ptr(theta)),
                    "gpry_set_theta")

    def set_affine(self, x_lo=None, x_span=None, y_mean=0.0, y_std=1.0, clip_hi=np.inf):
        tf = make_affine(x_lo, x_span, y_mean, y_std, clip_hi)
        self._check(self._lib.gpry_set_affine(self._h, C.byref(tf)), "gpry_set_affine")

    # -- kernels --------------------------------------------------------------------
    def kernel_train(self, add_alpha=False):
        K = np.empty((self.N, self.N))
        self._check(self._lib.gpry_kernel_train(self._h, int(bool(add_alpha)), _ptr(K)),
                    "gpry_kernel_train")
        return K

    def kernel_cross(self, Xc_):
        Xc_ = _f64(Xc_)
        M = Xc_.shape[0]
        K = np.empty((M, self.N))
        if M:
            self._check(self._lib.gpry_kernel_cross(self._h, _ptr(Xc_), M, _ptr(K)),
                        "gpry_kernel_cross")
        return K

    # -- factor / lml ---------------------------------------------------------------
    def factorize(self):
        info = C.c_int(0)
        self._check(self._lib.gpry_factorize(self._h, C.byref(info)), "gpry_factorize")
        return info.value

    def append_rows(self, Xnew_, ynew_, alphanew):
        """Extend the factor by the rows ``Xnew_`` (bordered update, fixed theta).  Returns ``info``."""
        Xnew_ = _f64(np.atleast_2d(Xnew_))
        k = Xnew_.shape[0]
        if Xnew_.shape[1] != self.d:
            raise ValueError(f"expected {self.d} columns, got {Xnew_.shape[1]}")
        ynew_ = _f64(ynew_, (k,))
        alphanew = _f64(np.broadcast_to(alphanew, (k,)))
        info = C.c_int(0)
        self._check(self._lib.gpry_append_rows(self._h, _ptr(Xnew_), _ptr(ynew_), _ptr(alphanew), k,
                                               C.byref(info)), "gpry_append_rows")
        if info.value == 0:
            self.N += k
        return info.value

    def get_factor(self, want_L=True, want_V=True, want_alpha=True):
        N = self.N
        L = np.empty((N, N)) if want_L else None
        V = np.empty((N, N)) if want_V else None
        a = np.empty(N) if want_alpha else None
        self._check(self._lib.gpry_get_factor(self._h, _ptr(L), _ptr(V), _ptr(a)),
                    "gpry_get_factor")
        return L, V, a

    def lml(self, theta, eval_gradient=False):
        # persistent argument buffers: at small N an evaluation takes 30-60 us, of which building two arrays and four
        # ctypes pointers per call used to be 3-4
        buf = getattr(self, "_lml_buf", None)
        if buf is None or len(buf[0]) != self.d + 1:
            th, gr = np.zeros(self.d + 1), np.zeros(self.d + 1)
            val, info = C.c_double(0.0), C.c_int(0)
            buf = self._lml_buf = (th, gr, C.c_void_p(th.ctypes.data), C.c_void_p(gr.ctypes.data), val, info,
                                   C.byref(val), C.byref(info))
        th, gr, pth, pgr, val, info, rval, rinfo = buf
        th[:] = theta                          # (raises on a shape mismatch)
        rc = self._lib.gpry_lml(self._h, pth, 1 if eval_gradient else 0, rval, pgr, rinfo)
        if rc != 0:
            self._check(rc, "gpry_lml")
        if eval_gradient:
            return val.value, gr.copy(), info.value
        return val.value, info.value

    def lml_batch(self, thetas, eval_gradient=True):
        """``lml`` for every row of ``thetas`` in one call: ``(lml (B,), grad (B, 1 + d), info (B,))`` (``(lml, info)``
        without gradients).  N <= 128, d <= 16: one launch, one workgroup per theta, the bits of single ``lml`` calls."""
        thetas = _f64(np.atleast_2d(thetas))
        B, w = thetas.shape
        if w != self.d + 1:
            raise ValueError(f"expected {self.d + 1} columns, got {w}")
        lml = np.empty(B)
        grad = np.zeros((B, w)) if eval_gradient else None
        info = np.zeros(B, dtype=np.int32)
        self._check(self._lib.gpry_lml_batch(self._h, _ptr(thetas), B, int(bool(eval_gradient)), _ptr(lml), _ptr(grad),
                                             _ptr(info)), "gpry_lml_batch")
        return (lml, grad, info) if eval_gradient else (lml, info)

    # -- predict / sweep ------------------------------------------------------------
    def predict(self, X, return_std=False, mask=None):
        if not return_std and mask is None and 0 < len(X) <= 8:
            # the point-by-point callers (nested samplers, MCMC): persistent buffers and cached pointers -- building
            # four ctypes pointers and two arrays per call cost more than the device round trip
            fast = getattr(self, "_small", None)
            if fast is None or fast[0].shape[1] != self.d:
                xin, out = np.zeros((8, max(self.d, 1))), np.zeros(8)
                fast = self._small = (xin, out, C.c_void_p(xin.ctypes.data), C.c_void_p(out.ctypes.data))
            xin, out, pin, pout = fast
            M = len(X)
            if np.ndim(X) != 2 or np.shape(X)[1] != self.d:      # (an assignment would broadcast a vector or a column)
                raise ValueError(f"expected an (M, {self.d}) array, got shape {np.shape(X)}")
            xin[:M] = X            # (converts dtype / strides)
            rc = self._lib.gpry_predict(self._h, pin, M, None, pout, None)
            if rc != 0:
                self._check(rc, "gpry_predict")
            return out[:M].copy()
        X = _f64(X)
        M = X.shape[0]
        if X.ndim != 2 or X.shape[1] != self.d:
            raise ValueError(f"expected an array of shape (M, {self.d}), got {X.shape}")
        mean = np.empty(M)
        std = np.empty(M) if return_std else None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if M:
            self._check(self._lib.gpry_predict(self._h, _ptr(X), M, _ptr(mask), _ptr(mean),
                                               _ptr(std)), "gpry_predict")
        return (mean, std) if return_std else mean

    def serve_stats(self):
        """(launches, requests) of the resident predict kernel that answers small mean-only ``predict`` calls."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.gpry_debug_serve_stats(self._h, C.byref(a), C.byref(b)), "gpry_debug_serve_stats")
        return a.value, b.value

    def set_gates(self, sv=None, coef=None, gamma=0.0, intercept=0.0, positive_is_finite=True,
                  trust_bounds=None):
        """SVM / trust-region verdicts computed by the device inside ``sweep_logexp``."""
        n_sv = 0
        if sv is not None and len(sv):
            sv = _f64(sv)
            n_sv = sv.shape[0]
            coef = _f64(coef, (n_sv,))
        else:
            sv = coef = None
        tb = None if trust_bounds is None else _f64(trust_bounds, (self.d, 2))
        self._check(self._lib.gpry_set_gates(self._h, _ptr(sv), _ptr(coef), n_sv, float(gamma),
                                             float(intercept), int(bool(positive_is_finite)), _ptr(tb)),
                    "gpry_set_gates")

    def predict_grad(self, x, want_kinv=True, want_kgrad=False, want_mean=True):
        """x-gradient contractions for one point: ``(G^T alpha_, G^T K^-1 k*[, G])``."""
        x = _f64(x, (self.d,))
        mg = np.empty(self.d) if want_mean else None
        kg = np.empty(self.d)
        G = np.empty((self.N, self.d)) if want_kgrad else None
        self._check(self._lib.gpry_predict_grad(self._h, _ptr(x), int(bool(want_kinv)), _ptr(G),
                                                _ptr(mg), _ptr(kg)), "gpry_predict_grad")
        return (mg, kg, G) if want_kgrad else (mg, kg)

    def predict_point(self, x, mask_bits=0, want_kinv=True):
        """``(mean, std, G^T alpha_, G^T K^-1 k*, verdict)`` of ONE point in one call (mean / std finalised as by ``predict``;
        ``verdict``: the mask bits that were applied -- ``mask_bits`` and, with option "predict_gates", the device's own gates).
        Called once per step of an acquisition optimiser: argument buffers and their pointers are kept."""
        buf = getattr(self, "_point_buf", None)
        if buf is None or len(buf[0]) != self.d:
            xin, mg, kg = np.zeros(self.d), np.zeros(self.d), np.zeros(self.d)
            mean, std, bits = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
            buf = self._point_buf = (xin, mg, kg, mean, std, bits, C.c_void_p(xin.ctypes.data), C.byref(mean), C.byref(std),
                                     C.c_void_p(mg.ctypes.data), C.c_void_p(kg.ctypes.data), C.byref(bits))
        xin, mg, kg, mean, std, bits, px, pmean, pstd, pmg, pkg, pbits = buf
        xin[:] = x                                 # (raises on a shape mismatch; converts dtype / strides)
        rc = self._lib.gpry_predict_point(self._h, px, int(mask_bits), 1 if want_kinv else 0, pmean, pstd, pmg, pkg, pbits)
        if rc != 0:
            self._check(rc, "gpry_predict_point")
        return mean.value, std.value, mg.copy(), kg.copy(), bits.value

    def predict_grad_batch(self, X, want_kinv=True):
        """``(mean, std, G^T alpha_, G^T K^-1 k*)`` for every row of ``X`` in one call."""
        X = _f64(np.atleast_2d(X))
        m = X.shape[0]
        mean, std = np.empty(m), np.empty(m)
        mg = np.empty((m, self.d))
        kg = np.zeros((m, self.d))
        if m:
            self._check(self._lib.gpry_predict_grad_batch(self._h, _ptr(X), m, int(bool(want_kinv)), _ptr(mean),
                                                          _ptr(std), _ptr(mg), _ptr(kg)), "gpry_predict_grad_batch")
        return mean, std, mg, kg

    def sweep_logexp(self, X, zeta, baseline, sigma_n, mask=None, M=None, want=("y", "sigma", "acq")):
        """Run the fused sweep.  ``X=None`` re-uses the candidate set resident on the device."""
        if X is not None:
            X = _f64(X)
            M = X.shape[0]
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        out = {k: (np.empty(M) if k in want else None) for k in ("y", "sigma", "acq")}
        n_nan = C.c_int64(0)
        self._check(self._lib.gpry_sweep_logexp(
            self._h, _ptr(X), M, _ptr(mask), float(zeta), float(baseline), float(sigma_n),
            _ptr(out["y"]), _ptr(out["sigma"]), _ptr(out["acq"]), C.byref(n_nan)),
            "gpry_sweep_logexp")
        out["n_nan"] = n_nan.value
        self.sweep_epoch = getattr(self, "sweep_epoch", 0) + 1    # the resident arrays changed
        self._sweep_M = M
        return out

    PANEL_FORMS = {0: "none", 1: "mfma", 2: "difference", 3: "small", 4: "hybrid"}

    def sweep_info(self):
        """How the cross-kernel panel of the last sweep / panel predict was built (``gpry_sweep_info``) and the error
        estimates of the model that decided it."""
        form = C.c_int(0)
        est = np.zeros(4)
        self._check(self._lib.gpry_sweep_info(self._h, C.byref(form), _ptr(est)), "gpry_sweep_info")
        return {"panel_form": self.PANEL_FORMS.get(form.value, str(form.value)), "panel_error_estimate": float(est[0]),
                "panel_error_mean_worst_case": float(est[1]), "panel_error_variance": float(est[2]), "panel_gate": float(est[3])}

    def sweep_fetch(self, want=("y", "sigma")):
        """Arrays of the last sweep that are still resident on the device."""
        M = self._sweep_M
        out = {k: (np.empty(M) if k in want else None) for k in ("y", "sigma", "acq")}
        self._check(self._lib.gpry_sweep_fetch(self._h, M, _ptr(out["y"]), _ptr(out["sigma"]),
                                               _ptr(out["acq"])), "gpry_sweep_fetch")
        return out

    def sweep_topk(self, Kp, exclude=None):
        Kp = int(Kp)
        top = np.zeros(max(Kp, 1), dtype=CAND_DTYPE)
        n_out, bound = C.c_int64(0), C.c_double(0.0)
        if exclude is not None and len(exclude):
            exclude = np.ascontiguousarray(np.sort(np.asarray(exclude, dtype=np.int64)))
            nex = len(exclude)
        else:
            exclude, nex = None, 0
        self._check(self._lib.gpry_sweep_topk(self._h, Kp, _ptr(exclude), nex, _ptr(top),
                                              C.byref(n_out), C.byref(bound)), "gpry_sweep_topk")
        return top[:n_out.value], bound.value

    # -- Kriging believer -----------------------------------------------------------
    def kb_reset(self):
        self._check(self._lib.gpry_kb_reset(self._h), "gpry_kb_reset")

    def kb_register(self, X, want_var0=True):
        X = _f64(X)
        m = X.shape[0]
        first = C.c_int64(0)
        var0 = np.empty(m) if want_var0 else None
        self._check(self._lib.gpry_kb_register(self._h, _ptr(X), m, C.byref(first), _ptr(var0)),
                    "gpry_kb_register")
        return first.value, var0

    def kb_gram(self, p, n):
        G, kv = np.empty(n), np.empty(n)
        nn = C.c_int64(0)
        self._check(self._lib.gpry_kb_gram(self._h, int(p), _ptr(G), _ptr(kv), C.byref(nn)),
                    "gpry_kb_gram")
        if nn.value != n:
            raise GpryHipError(f"kb_gram: device holds {nn.value} candidates, host expected {n}")
        return G, kv

    # -- measurement ----------------------------------------------------------------
    def timing_reset(self):
        self._check(self._lib.gpry_timing_reset(self._h), "gpry_timing_reset")

    def timing(self, name):
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        self._lib.gpry_timing_get(self._h, name.encode(), C.byref(ms), C.byref(cnt))
        return ms.value, cnt.value

    def debug_gemm(self, A, B, C0, M, N, K, a_trans=False, b_trans=False, epi=0, kmode=0,
                   lower_only=False, tile_map=0):
        """Unit-test hook for the MFMA GEMM engine (see include/gpry_hip.h)."""
        A, B = _f64(A), _f64(B)
        crow = (M + 127) // 128 if epi == 3 else M
        Cout = np.ascontiguousarray(np.zeros((crow, N)) if C0 is None else C0, dtype=np.float64).copy()
        self._check(self._lib.gpry_debug_gemm(self._h, _ptr(A), _ptr(B), _ptr(Cout), M, N, K,
                                              int(a_trans), int(b_trans), int(epi), int(kmode),
                                              int(lower_only), int(tile_map)), "gpry_debug_gemm")
        return Cout

    def debug_logexp(self, mu, sigma, zeta, baseline, sigma_n):
        """The sweep's acquisition epilogue on given (mean, std) pairs (test hook)."""
        mu, sigma = _f64(mu), _f64(sigma)
        acq = np.empty(len(mu))
        self._check(self._lib.gpry_debug_logexp(self._h, _ptr(mu), _ptr(sigma), len(mu), float(zeta),
                                                float(baseline), float(sigma_n), _ptr(acq)), "gpry_debug_logexp")
        return acq


    def microbench(self, kind, nbytes=0):
        v = C.c_double(0.0)
        self._check(self._lib.gpry_microbench(self._h, int(kind), int(nbytes), C.byref(v)),
                    "gpry_microbench")
        return v.value


class DeviceGroup:
    """Several contexts behind one call (``gpry_group_*``): the sharded sweep of ONE process.

    ``devices``: device indices, repeats allowed (several contexts on one GPU).  ``adopt``: a
    ``Device`` on ``devices[0]`` that becomes member 0 and keeps belonging to its owner (the
    regressor's own context: its factor is used as it is, the other members get the model through
    ``set_model``).  The methods mirror ``Device``'s sweep calls on the whole pool."""

    def __init__(self, devices, adopt=None):
        self._lib = load_library()
        devices = [int(v) for v in devices]
        if not devices:
            raise ValueError("DeviceGroup needs at least one device")
        self.devices = devices
        self._adopt = adopt             # keeps the adopted context alive as long as the group
        arr = (C.c_int * len(devices))(*devices)
        self._h = C.c_void_p()
        rc = self._lib.gpry_group_create(len(devices), arr, adopt._h if adopt is not None else None,
                                         C.byref(self._h))
        if rc != 0:
            msg = self._lib.gpry_group_last_error(None).decode(errors="replace")
            self._h = None
            raise GpryHipError(f"gpry_group_create({devices}) failed ({rc}): {msg}")
        n, tr = C.c_int(0), C.c_int(0)
        self._lib.gpry_group_size(self._h, C.byref(n), C.byref(tr))
        self.size, self.transport = n.value, {0: "host", 1: "rccl"}[tr.value]
        self.note = self._lib.gpry_group_last_error(self._h).decode(errors="replace")
        self.sweep_epoch = 0
        self._sweep_M = 0
        self.d = 0

    def _check(self, rc, what):
        if rc != 0:
            msg = self._lib.gpry_group_last_error(self._h).decode(errors="replace")
            raise GpryHipError(f"{what} failed ({rc}): {msg}")

    def close(self):
        if getattr(self, "_h", None):
            try:
                self._lib.gpry_group_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def __del__(self):
        self.close()

    def member(self, i):
        """Member ``i`` as a (non-owning) ``Device`` -- options, timers."""
        h = self._lib.gpry_group_member(self._h, int(i))
        if not h:
            raise IndexError(i)
        v = Device._view(h, self.devices[i])
        v._group = self                # the view must not outlive the group
        return v

    def set_model(self, X_, y_, alpha, kernel_id, theta, affine=None):
        """Replicate (training set, theta, affine maps) on every owned member and factorise there."""
        X_ = _f64(X_)
        N, d = X_.shape
        y_ = _f64(y_, (N,))
        alpha = _f64(np.broadcast_to(alpha, (N,)))
        theta = _f64(theta, (d + 1,))
        tf = make_affine(*affine) if affine is not None else None
        info = C.c_int(0)
        self._check(self._lib.gpry_group_set_model(self._h, _ptr(X_), _ptr(y_), _ptr(alpha), N, d, int(kernel_id),
                                                   _ptr(theta), C.byref(tf) if tf is not None else None,
                                                   C.byref(info)), "gpry_group_set_model")
        self.d = d
        return info.value

    def set_gates(self, sv=None, coef=None, gamma=0.0, intercept=0.0, positive_is_finite=True,
                  trust_bounds=None):
        n_sv = 0
        if sv is not None and len(sv):
            sv = _f64(sv)
            n_sv = sv.shape[0]
            coef = _f64(coef, (n_sv,))
        else:
            sv = coef = None
        tb = None if trust_bounds is None else _f64(trust_bounds)
        self._check(self._lib.gpry_group_set_gates(self._h, _ptr(sv), _ptr(coef), n_sv, float(gamma),
                                                   float(intercept), int(bool(positive_is_finite)), _ptr(tb)),
                    "gpry_group_set_gates")

    def sweep_logexp(self, X, zeta, baseline, sigma_n, mask=None, M=None, want=("y", "sigma", "acq")):
        if X is not None:
            X = _f64(X)
            M = X.shape[0]
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
        out = {k: (np.empty(M) if k in want else None) for k in ("y", "sigma", "acq")}
        n_nan = C.c_int64(0)
        self._check(self._lib.gpry_group_sweep_logexp(
            self._h, _ptr(X), M, _ptr(mask), float(zeta), float(baseline), float(sigma_n),
            _ptr(out["y"]), _ptr(out["sigma"]), _ptr(out["acq"]), C.byref(n_nan)), "gpry_group_sweep_logexp")
        out["n_nan"] = n_nan.value
        self.sweep_epoch += 1
        self._sweep_M = M
        return out

    def sweep_fetch(self, want=("y", "sigma")):
        M = self._sweep_M
        out = {k: (np.empty(M) if k in want else None) for k in ("y", "sigma", "acq")}
        self._check(self._lib.gpry_group_sweep_fetch(self._h, M, _ptr(out["y"]), _ptr(out["sigma"]),
                                                     _ptr(out["acq"])), "gpry_group_sweep_fetch")
        return out

    def sweep_topk(self, Kp, exclude=None):
        """(records sorted by (acq desc, global idx desc), bound, exhausted)."""
        Kp = int(Kp)
        top = np.zeros(max(Kp, 1) * self.size, dtype=CAND_DTYPE)
        n_out, bound, exh = C.c_int64(0), C.c_double(0.0), C.c_int(0)
        if exclude is not None and len(exclude):
            exclude = np.ascontiguousarray(np.sort(np.asarray(exclude, dtype=np.int64)))
            nex = len(exclude)
        else:
            exclude, nex = None, 0
        self._check(self._lib.gpry_group_sweep_topk(self._h, Kp, _ptr(exclude), nex, _ptr(top), C.byref(n_out),
                                                    C.byref(bound), C.byref(exh)), "gpry_group_sweep_topk")
        return top[:n_out.value], bound.value, bool(exh.value)

    def lml_batch(self, thetas, eval_gradient=True):
        thetas = _f64(np.atleast_2d(thetas))
        n, w = thetas.shape
        lml = np.empty(n)
        grad = np.zeros((n, w)) if eval_gradient else None
        info = np.zeros(n, dtype=np.int32)
        self._check(self._lib.gpry_group_lml_batch(self._h, _ptr(thetas), n, int(bool(eval_gradient)), _ptr(lml),
                                                   _ptr(grad), _ptr(info)), "gpry_group_lml_batch")
        return (lml, grad, info) if eval_gradient else (lml, info)


class RcclComm:
    """RCCL communicator of one rank (one process per GPU)."""

    def __init__(self, dev, world, rank, unique_id):
        self._dev = dev
        self._lib = dev._lib
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        uid = np.frombuffer(bytes(unique_id), dtype=np.uint8).copy()
        dev._check(self._lib.gpry_comm_init(dev._h, self.world, self.rank, _ptr(uid),
                                            C.byref(self._h)), "gpry_comm_init")

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = np.zeros(128, dtype=np.uint8)
        if lib.gpry_comm_unique_id(_ptr(buf)) != 0:
            raise GpryHipError("gpry_comm_unique_id failed: " +
                               lib.gpry_last_error(None).decode(errors="replace"))
        return buf.tobytes()

    def info(self):
        """(ranks, rank, device) as RCCL reports them for this communicator."""
        n, r, dv = C.c_int(0), C.c_int(0), C.c_int(0)
        self._dev._check(self._lib.gpry_comm_info(self._h, C.byref(n), C.byref(r), C.byref(dv)),
                         "gpry_comm_info")
        return n.value, r.value, dv.value

    def allgather(self, arr):
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.world,) + arr.shape, dtype=arr.dtype)
        self._dev._check(self._lib.gpry_comm_allgather(self._h, _ptr(arr), arr.nbytes, _ptr(out)),
                         "gpry_comm_allgather")
        return out

    def allreduce_max(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).copy()
        self._dev._check(self._lib.gpry_comm_allreduce_max(self._h, _ptr(arr), arr.size),
                         "gpry_comm_allreduce_max")
        return arr

    def barrier(self):
        self._dev._check(self._lib.gpry_comm_barrier(self._h), "gpry_comm_barrier")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gpry_comm_destroy(self._h)
            self._h = None
