"""Device-backed Gaussian-process regressor with GPry's incremental-data interface.

Host mirror of ``gpry.gpr.GaussianProcessRegressor`` (gpry/gpr.py:27): the object model,
bookkeeping and control flow stay in Python (``append_to_data`` :577-753,
``fit_gpr_hyperparameters`` :883-994 with scipy's L-BFGS-B, ``_update_model`` :996-1020,
``predict`` :1022-1273, ``predict_std`` :1275-1352, ``__deepcopy__`` :1354-1433); every
O(N^2)...O(M N^2) array operation is a call into ``libgpry_hip.so``:

    kernel build + Cholesky + L^-1 + alpha_   -> gpry_factorize
    log marginal likelihood (+ gradient)      -> gpry_lml
    posterior mean / std with the affine maps -> gpry_predict / gpry_sweep_logexp
    conditioned models of the KB ranking      -> gpry_kb_*  (bordered factor)

There is no CPU path: without the library or without a GPU the methods raise.
"""
import copy
import os
import sys
import warnings
from collections.abc import Mapping
from numbers import Number

import numpy as np
import scipy.optimize

from gpry_amd import _lib
from gpry_amd.kernels import RBF, Matern, ConstantKernel as C, clone
from gpry_amd.preprocessing import DummyPreprocessor, Normalize_bounds, Normalize_y
from gpry_amd.svm import SVM
from gpry_amd.tools import (check_random_state, delta_logp_of_1d_nstd, generic_params_names,
                            get_Xnumber, is_in_bounds, shrink_bounds)

try:  # estimator tags so that sklearn.base.is_regressor() accepts the object
    from sklearn.base import BaseEstimator as _BE, RegressorMixin as _RM
except Exception:  # pragma: no cover
    _BE = _RM = object


# One process per GPU is announced differently by every launcher: torchrun sets WORLD_SIZE / LOCAL_RANK; the reference's
# own parallel mode is mpi4py under mpirun / srun (gpry/mpi.py:18-28, gpry/run.py:1254-1275), which set these instead.
# A world size alone does not make a process a rank: inside an `sbatch --ntasks=8` allocation a plain `python run.py`
# (no srun) sees SLURM_NTASKS=8 as well and must keep all its GPUs -- the launcher's rank variable has to be there too.
# Slurm: the task count of the STEP (SLURM_STEP_NUM_TASKS, set by srun for its own tasks), not of the allocation -- a single
# `python run.py` inside an interactive step (`salloc` with use_interactive_step, `srun --pty bash`) has SLURM_STEP_ID and
# SLURM_NTASKS=8 as well, but a step of ONE task, and keeps every GPU.
_LAUNCHERS = (("WORLD_SIZE", ("RANK", "LOCAL_RANK")),
              ("OMPI_COMM_WORLD_SIZE", ("OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_LOCAL_RANK")),
              ("PMI_SIZE", ("PMI_RANK", "MPI_LOCALRANKID")),
              ("SLURM_STEP_NUM_TASKS", ("SLURM_PROCID",)))
_WORLD_SIZE_VARS = tuple(v for v, _ in _LAUNCHERS)
_LOCAL_RANK_VARS = ("LOCAL_RANK", "OMPI_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID", "SLURM_LOCALID")
_LAUNCH_NOTE = {"said": False}


def multi_process_launch():
    """True if this process is one rank of a multi-process launch: a launcher's world-size variable > 1 TOGETHER WITH that
    launcher's rank variable, or an initialised mpi4py world of more than one rank.  A world size without a rank (a
    single process inside a multi-task allocation) is not one; that decision is logged once."""
    for size_var, rank_vars in _LAUNCHERS:
        try:
            many = int(os.environ.get(size_var, "1")) > 1
        except ValueError:
            many = False
        if not many:
            continue
        if any(os.environ.get(v, "") != "" for v in rank_vars):
            return True
        if not _LAUNCH_NOTE["said"]:
            _LAUNCH_NOTE["said"] = True
            import logging
            logging.getLogger("gpry_amd").info(
                "%s=%s without %s: treated as ONE process that may use every visible GPU, not as a rank of a "
                "multi-process launch", size_var, os.environ.get(size_var), " / ".join(rank_vars))
    mpi = sys.modules.get("mpi4py.MPI")
    try:
        return mpi is not None and mpi.Is_initialized() and mpi.COMM_WORLD.Get_size() > 1
    except Exception:
        return False


# options of a device context that change the arithmetic or the limits of an objective evaluation: extra contexts of a fit
# (thread farm, side-by-side groups) take the values of the model's own context, so that every context evaluates alike
# whatever the caller has set ("same bits as the sequential loop")
_FIT_CONTEXT_OPTIONS = ("chol", "chol_overlap", "chol_stacked", "factor_pipeline", "factor_pipeline_min", "gemm_dma", "gemm_streamk",
                        "gemm_small", "lml_small", "lml_cache", "lml_batch", "lml_batch_mb", "lml_schedule", "lml_streams",
                        "tp_block", "tp_tail")


def copy_fit_options(src, dst):
    if not (hasattr(src, "get_option") and hasattr(dst, "set_option")):
        return
    for key in _FIT_CONTEXT_OPTIONS:
        try:
            v = src.get_option(key)
            if dst.get_option(key) != v:
                dst.set_option(key, v)
        except Exception:       # an option this build does not know
            pass


def default_device_index():
    """GPU of this process: ``GPRY_HIP_DEVICE`` or the launcher's local rank (``LOCAL_RANK``,
    ``OMPI_COMM_WORLD_LOCAL_RANK``, ``MPI_LOCALRANKID``, ``SLURM_LOCALID``: one process per GPU), else 0.

    An index beyond the visible devices raises: two ranks silently sharing a GPU is what made the
    RCCL bootstrap fail with "invalid usage" (a communicator cannot hold one device twice).
    ``GPRY_HIP_DEVICE_WRAP=1`` restores the wrap-around for development boxes with fewer GPUs
    than ranks (no RCCL communicator can be built there)."""
    for var in ("GPRY_HIP_DEVICE",) + _LOCAL_RANK_VARS:
        if os.environ.get(var, "") != "":
            n, idx = _lib.device_count(), int(os.environ[var])
            if n > 0 and idx >= n:
                if n == 1 and any(os.environ.get(v, "") != "" for v in
                                  ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")):
                    return 0        # the launcher already gave this process a GPU of its own
                if os.environ.get("GPRY_HIP_DEVICE_WRAP", "") == "1":
                    return idx % n
                raise _lib.GpryHipError(
                    f"{var}={idx} but only {n} GPU(s) are visible: one process per GPU "
                    "(set GPRY_HIP_DEVICE_WRAP=1 to share devices, without RCCL)")
            return idx
    return 0


def fit_contexts():
    """Device contexts (= host threads) that share the optimiser restarts of one fit:
    ``GPRY_HIP_FIT_CONTEXTS``, default 3; 1 = the reference's sequential loop."""
    try:
        return max(1, int(os.environ.get("GPRY_HIP_FIT_CONTEXTS", "3")))
    except ValueError:
        return 3


_BATCH_GROUP_MIN_RUNS = 3         # a group of a side-by-side fit holds at least this many runs


def batch_contexts(n_train=None):
    """Independent groups (device contexts, host threads) that the runs of a side-by-side fit are dealt out over:
    ``GPRY_HIP_FIT_BATCH_CONTEXTS`` if set, else by the size of the training set (measured, tools/r04/time_fit_crossover.py)."""
    env = os.environ.get("GPRY_HIP_FIT_BATCH_CONTEXTS", "")
    if env != "":
        try:
            return max(1, int(env))
        except ValueError:
            pass
    if n_train is None:
        return 2
    # measured on one MI355X (26 restarts, d = 8): N = 200 60 / 60 / 60 ms with 1 / 2 / 3 groups, 400: 61 / 57 / 55, 800: 124 /
    # 113 / 103, 1600: 310 / 278 / 257, 4096: 2302 / 2140 / 2157; four and six groups are slower everywhere
    return 1 if n_train <= 300 else 3


def fit_schedule(n_train, n_runs):
    """Schedule of the batched objective for a side-by-side fit of ``n_runs`` optimiser runs on ``n_train`` points:
    ``(name, groups)`` with name "latency" (every theta gets the launches and the bits of a single evaluation; the runs dealt
    over ``batch_contexts`` independent groups) or "throughput" (``gpry_hip.h``: option "lml_schedule" = 1 -- whole-tile
    products, the Cholesky in column blocks, stream groups inside a call; a theta's value does not depend on how many thetas
    share the call, and differs from the latency schedule's by rounding).  ``GPRY_HIP_FIT_SCHEDULE`` = latency / throughput
    forces one; by default the throughput schedule is taken where whole fits were measured ahead (one MI355X, tools/r06/time_fit.py,
    profiles/r06_tp.md): 42 restarts at N = 4096, d = 16 in 2.4 s against 3.2 s, 0.50 against 0.58 s at N = 2304; equal at N = 2048 (0.39 s), behind at N = 1024
    (110 against 100 ms) -- per call it is ahead from 4 thetas at N = 4096, 10 at 2048 and 24 at 1024, and a fit's rounds thin
    out as its runs converge.  So: from 2304 padded rows on, with at least six runs.  ``GPRY_HIP_FIT_TP_GROUPS`` (default 1): groups of a throughput fit -- its calls overlap their
    own stream groups, so all runs share one context and every round is as wide as the fit allows."""
    env = os.environ.get("GPRY_HIP_FIT_SCHEDULE", "auto").lower()
    try:
        tp_groups = max(1, int(os.environ.get("GPRY_HIP_FIT_TP_GROUPS", "1")))
    except ValueError:
        tp_groups = 1
    if env == "latency":
        return "latency", None
    if env != "throughput":
        npad = -(-int(n_train) // 128) * 128
        if npad < 2304 or n_runs < 6:
            return "latency", None
    return "throughput", tp_groups


def fit_context_devices(own, n_restarts, spec=None):
    """Device index of every context that shares the optimiser restarts of one fit in THIS process; entry 0
    is the model's own context.

    ``spec`` (``GaussianProcessRegressor.fit_devices``): explicit list, one entry per context, repeats
    allowed.  ``None``: ``GPRY_HIP_DEVICES`` (a list of GPUs, or ``all`` / ``none``), else every visible GPU
    unless this process is one rank of a multi-process launch (``multi_process_launch``: torchrun's ``WORLD_SIZE``, or
    ``OMPI_COMM_WORLD_SIZE`` / ``PMI_SIZE`` / ``SLURM_STEP_NUM_TASKS`` of the reference's mpirun / srun mode -- one process per
    GPU, the ranks farm the restarts among themselves, ``gpry_amd.parallel`` / ``gpry/run.py:1254-1275``); ``fit_contexts()`` contexts per GPU,
    dealt out round-robin so that the first restarts land on distinct GPUs.  An unmodified single-process
    ``gpry.Runner`` on an 8-GPU node (``gpry/run.py:315-325``: without mpi4py there is one rank) thereby
    spreads its 10+2d restarts over all GPUs instead of running them on GPU 0."""
    per = fit_contexts()
    if spec is None:
        env = os.environ.get("GPRY_HIP_DEVICES", "").strip().lower()
        if env not in ("", "none", "1", "all"):
            devs = [int(v) for v in env.split(",")]
        elif env in ("none", "1") or (env == "" and multi_process_launch()):
            devs = [own]
        else:
            n = _lib.device_count()
            devs = list(range(n)) if n > 1 else [own]
        devs = [own] + [v for v in dict.fromkeys(devs) if v != own]
        spec = [devs[i % len(devs)] for i in range(per * len(devs))]
    else:
        spec = [own] + [int(v) for v in list(spec)[1:]]
    return spec[:max(1, min(len(spec), int(n_restarts)))]


_SCRATCH = None


def _scratch_device():
    """Shared context for stand-alone kernel evaluations (``kernel_(X, Y)``)."""
    global _SCRATCH
    if _SCRATCH is None:
        _SCRATCH = _lib.Device(default_device_index())
    return _SCRATCH


class GaussianProcessRegressor(_RM, _BE):
    """See the module docstring; constructor arguments as gpry/gpr.py:265-271."""

    def __init__(self, kernel="RBF", output_scale_prior=[1e-2, 1e3],
                 length_scale_prior=[1e-3, 1e1], noise_level=1e-2, clip_factor=1.1,
                 optimizer="fmin_l_bfgs_b", n_restarts_optimizer=0,
                 preprocessing_X=None, preprocessing_y=None,
                 account_for_inf="SVM", inf_threshold="20s", keep_min_finite=None,
                 trust_region_factor=None, trust_region_nstd=None,
                 bounds=None, random_state=None, verbose=1):
        self.output_scale_prior = output_scale_prior
        self.length_scale_prior = length_scale_prior
        self.account_for_inf = account_for_inf
        self.n_last_appended = 0
        self.n_last_appended_finite = 0
        self.newly_appended_for_inv = 0
        self.preprocessing_X = DummyPreprocessor if preprocessing_X is None else preprocessing_X
        self.preprocessing_y = DummyPreprocessor if preprocessing_y is None else preprocessing_y
        self.noise_level = noise_level
        if clip_factor is not None and clip_factor < 1:
            raise ValueError("'clip_factor' must be >= 1, or None for no clippling.")
        self.clip_factor = clip_factor
        self.n_eval = 0
        self.n_eval_loglike = 0
        self.verbose = verbose
        self.inf_value = np.inf
        self.minus_inf_value = -np.inf
        self._fitted = False
        self.bounds = None if bounds is None else np.asarray(bounds, dtype=float)
        self.trust_bounds = None
        self.trust_region_factor = trust_region_factor
        self.trust_region_nstd = trust_region_nstd
        self.inf_threshold = inf_threshold
        self.X_train = np.empty((0, 0))
        if keep_min_finite is None and self.bounds is None:
            self.keep_min_finite = 2
        else:
            self.keep_min_finite = keep_min_finite if keep_min_finite is not None else max(2, self.d)
        if isinstance(account_for_inf, str) and account_for_inf.lower() == "svm":
            self.infinities_classifier = SVM(random_state=random_state)
        elif account_for_inf is False:
            self.infinities_classifier = None
        else:
            self.infinities_classifier = account_for_inf
        if self.infinities_classifier is not None:
            if not getattr(self.preprocessing_y, "is_linear", False):
                warnings.warn("If using a standard classifier for infinities, the y-preprocessor "
                              "needs to be linear (declare an attr ``is_linear=True``). This may "
                              "lead to errors further in the pipeline.")
            if self.inf_threshold is None:
                raise ValueError("Specify 'inf_threshold' if using infinities classifier.")
            value, in_sigma, power = get_Xnumber(self.inf_threshold, "s", None, dtype=float,
                                                 varname="inf_threshold")
            if power is not None:
                raise ValueError("Power for sigma not supported.")
            self._diff_threshold = (self.compute_threshold_given_sigma(value, self.d)
                                    if in_sigma else value)
        if isinstance(kernel, str):
            kernel = {kernel: {}}
        if isinstance(kernel, Mapping):
            if len(kernel) != 1:
                raise ValueError("'kernel' must be a single-key dict.")
            name = list(kernel)[0]
            kargs = kernel[name] or {}
            if self.bounds is None:
                raise ValueError(f"You selected used the automatically constructed '{name}' kernel "
                                 "without specifying prior bounds.")
            self.bounds_ = self.preprocessing_X.transform_bounds(self.bounds)
            try:
                cls = {"rbf": RBF, "matern": Matern}[name.lower()]
            except KeyError as excpt:
                raise ValueError("Currently only 'RBF' and 'Matern' are supported as standard "
                                 f"kernels. Got '{name}'.") from excpt
            scale0 = np.sqrt(output_scale_prior[0] * output_scale_prior[1])
            length0 = np.sqrt(length_scale_prior[0] * length_scale_prior[1])
            kernel = (C(scale0 ** 2, [output_scale_prior[0] ** 2, output_scale_prior[1] ** 2]) *
                      cls([length0] * self.d, length_scale_prior, prior_bounds=self.bounds_, **kargs))
        self.kernel = kernel
        self.alpha = noise_level ** 2. if isinstance(noise_level, Number) else None
        self.optimizer = optimizer
        self.n_restarts_optimizer = n_restarts_optimizer
        self.random_state = random_state
        self.X_train, self.y_train = np.empty((0, self.d)), np.empty((0,))
        self.X_train_, self.y_train_ = None, None
        self.X_train_all, self.y_train_all = np.empty((0, self.d)), np.empty((0,))
        self.X_train_all_, self.y_train_all_ = None, None
        self.noise_level_ = None
        self.kernel_ = None
        self.log_marginal_likelihood_value_ = None
        self._init_device_state()

    # ---- device state ---------------------------------------------------------------
    def _init_device_state(self):
        self._dev = None
        self._dev_train_ok = False     # device holds X_train_, y_train_, alpha
        self._dev_factor_ok = False    # device factor matches kernel_.theta + training set
        self._host_factor = {}         # lazily fetched copies of L_, V_, alpha_
        self._kb = None                # Kriging-believer session on the current factor
        self._fit_devs = []            # extra contexts for concurrent optimiser restarts: (device index, context)
        self._dev_gates = None         # (key, on_device) of the gates the model's context holds (_sync_gates)

    @property
    def device(self):
        if self._dev is None:
            kind = getattr(self, "_device_kind", None) or _lib.Device
            self._dev = kind(default_device_index())
            if getattr(self._dev, "applies_gates_in_predict", False):
                self._dev.set_option("predict_gates", 1)     # predict() leaves classifier / trust box to the device
        return self._dev

    def _invalidate(self, train=False):
        if train:
            self._dev_train_ok = False
            self._train_epoch = getattr(self, "_train_epoch", 0) + 1     # pre-processors may have been refit
            self._affine_cache = None
        self._dev_factor_ok = False
        self._host_factor = {}
        self._kb = None

    def _upload_train(self):
        if not self._dev_train_ok:
            self.device.set_train(self.X_train_, self.y_train_, self.alpha)
            kern = self.kernel_ if self.kernel_ is not None else self.kernel
            self.device.set_theta(*kern.device_spec(self.d))   # fixes the kernel family too
            self._dev_train_ok = True
            self._dev_factor_ok = False

    def _x_affine(self):
        px = self.preprocessing_X
        if px is None or px is DummyPreprocessor or isinstance(px, DummyPreprocessor):
            return None, None
        if isinstance(px, Normalize_bounds) or (hasattr(px, "bounds_min") and hasattr(px, "bounds_max")):
            return np.asarray(px.bounds_min, float), np.asarray(px.bounds_max - px.bounds_min, float)
        raise NotImplementedError(f"X-preprocessor {type(px).__name__} has no device form (only "
                                  "Normalize_bounds / none are on the hot path; SURVEY.md section 2)")

    def _y_affine(self):
        py = self.preprocessing_y
        if py is None or py is DummyPreprocessor or isinstance(py, DummyPreprocessor):
            return 0.0, 1.0
        if isinstance(py, Normalize_y) or (hasattr(py, "mean_") and hasattr(py, "std_")):
            return float(py.mean_), float(py.std_)
        raise NotImplementedError(f"y-preprocessor {type(py).__name__} has no device form (only "
                                  "Normalize_y / none are on the hot path; SURVEY.md section 2)")

    def _clip_hi(self):
        if self.clip_factor is None:
            return np.inf
        # np.max / np.min: the builtin max() the reference uses walks the array in Python (50 us at
        # N = 1000 -- twice the whole device call of a one-point predict); same element either way
        return self.clip_factor * np.max(self.y_train) - (self.clip_factor - 1) * np.min(self.y_train)

    def _affine_args(self):
        """(x_lo, x_span, y_mean, y_std, clip_hi) of the fused pre-/post-processing; computed once per
        training set (the samplers call predict point by point)."""
        key = (self.clip_factor, id(self.preprocessing_X), id(self.preprocessing_y), len(self.y_train),
               getattr(self, "_train_epoch", 0))
        cached = getattr(self, "_affine_cache", None)
        if cached is None or cached[0] != key:
            lo, span = self._x_affine()
            mean_y, std_y = self._y_affine()
            cached = (key, (lo, span, mean_y, std_y, self._clip_hi()))
            self._affine_cache = cached
        return cached[1]

    def _push_affine(self):
        args = self._affine_args()
        if getattr(self, "_dev_affine", None) is not args or getattr(self, "_dev_affine_on", None) is not self.device:
            self.device.set_affine(*args)
            self._dev_affine, self._dev_affine_on = args, self.device

    def _device_theta(self, kernel=None):
        return (kernel or self.kernel_).device_spec(self.d)

    def _ensure_factor(self):
        """Bring the device factor up to date with (kernel_, training set)."""
        if self._dev_factor_ok:
            return
        self._upload_train()
        kid, theta = self._device_theta()
        self.device.set_theta(kid, theta)
        info = self.device.factorize()
        if info != 0:
            raise np.linalg.LinAlgError(
                "The kernel, %s, is not returning a positive definite matrix. Try gradually "
                "increasing the 'noise_level' parameter of your GaussianProcessRegressor "
                "estimator." % self.kernel_,
                f"{info}-th leading minor of the array is not positive definite")
        self._dev_factor_ok = True
        self._factor_epoch = getattr(self, "_factor_epoch", 0) + 1   # replicas elsewhere are stale now
        self._host_factor = {}
        self._kb = None

    def _fetch_factor(self, which):
        if which not in self._host_factor:
            self._ensure_factor()
            L, V, a = self.device.get_factor(want_L=(which == "L_"), want_V=(which == "V_"),
                                             want_alpha=(which == "alpha_"))
            self._host_factor[which] = {"L_": L, "V_": V, "alpha_": a}[which]
        return self._host_factor[which]

    # L_, V_, alpha_ live on the device; host copies are made on attribute access only
    @property
    def L_(self):
        return self._fetch_factor("L_")

    @property
    def V_(self):
        return self._fetch_factor("V_")

    @property
    def alpha_(self):
        return self._fetch_factor("alpha_")

    # ---- views of the training set (names and meaning as gpry/gpr.py:391-540) -------------------
    # "finite" = kept by the classifier and therefore in the GP; "all" = everything ever appended
    @property
    def d(self):
        return self.X_train.shape[1] if self.bounds is None else self.bounds.shape[0]

    def _finite_targets(self):
        return getattr(self, "y_train", ())

    n = property(lambda self: len(self._finite_targets()), doc="training points in the GP")
    n_finite = n
    fitted = property(lambda self: self._fitted)

    @property
    def y_max(self):
        return np.max(getattr(self, "y_train", (self.minus_inf_value,)))

    @property
    def n_total(self):
        clf = self.infinities_classifier
        return (clf.n if clf else 0) or self.n

    def _rejected(self, arr, empty_shape):
        clf = self.infinities_classifier
        return np.empty(shape=empty_shape) if clf is None else arr[~clf.y_finite]

    X_train_infinite = property(lambda self: self._rejected(self.X_train_all, (0, self.d)))
    y_train_infinite = property(lambda self: self._rejected(self.y_train_all, (0,)))

    @staticmethod
    def _tail(X, y, k):
        return X[-k:].copy(), y[-k:].copy()

    @property
    def last_appended_finite(self):
        return self._tail(self.X_train, self.y_train, self.n_last_appended_finite)

    @property
    def last_appended(self):
        if self.infinities_classifier is None:
            return self.last_appended_finite
        return self._tail(self.X_train_all, self.y_train_all, self.n_last_appended)

    @property
    def scales(self):
        """(output scale, length scales) in the units of the caller."""
        amplitude, lengths = np.sqrt(self.kernel_.k1.constant_value), np.atleast_1d(self.kernel_.k2.length_scale)
        return (self.preprocessing_y.inverse_transform_scale(amplitude),
                tuple(self.preprocessing_X.inverse_transform_scale(lengths)))

    def training_set_as_df(self):
        import pandas as pd
        y = self.y_train_all.copy()
        columns = {name: col for name, col in zip(generic_params_names(self.d), self.X_train_all.copy().T)}
        columns.update(y=y, is_finite=self.is_finite(y))
        return pd.DataFrame(columns)

    @property
    def abs_finite_threshold(self):
        return self.preprocessing_y.inverse_transform_scale(self.infinities_classifier.abs_threshold)

    def is_finite(self, y):
        """Threshold test on targets (no prediction involved)."""
        clf = self.infinities_classifier
        return np.ones(len(y), dtype=bool) if clf is None else clf.is_finite(self.preprocessing_y.transform(y))

    def predict_is_finite(self, X, validate=True):
        clf = self.infinities_classifier
        if clf is None:
            return np.ones(len(self.y_train_all), dtype=bool)      # (the reference's length, gpry/gpr.py:537)
        return clf.predict(np.ascontiguousarray(self.preprocessing_X.transform(X)), validate=validate)

    def set_random_state(self, random_state):
        self.random_state = random_state
        if self.infinities_classifier:
            self.infinities_classifier.random_state = check_random_state(random_state, convert_to_random_state=True)

    def update_trust_region(self):
        """Box around the training points within ``trust_region_nstd`` sigma of the best one (all of them without it),
        widened by ``trust_region_factor`` (gpry/gpr.py:554-575); the level is raised in steps of 0.1 sigma until it
        holds min(d, n) points."""
        if self.trust_region_factor is None:
            return
        inside = self.X_train
        if self.trust_region_nstd is not None:
            nstd, below_top = self.trust_region_nstd, max(self.y_train) - self.y_train
            inside = inside[:0]
            while len(inside) < min(self.d, self.n):
                inside = self.X_train[np.where(below_top < delta_logp_of_1d_nstd(nstd, self.d))]
                nstd = nstd + 0.1
        self.trust_bounds = shrink_bounds(self.bounds, inside, factor=self.trust_region_factor)

    # ---- data ---------------------------------------------------------------------------
    def append_to_data(self, X, y, noise_level=None, fit_gpr=True, fit_classifier=True):
        """Append points and update the model (gpry/gpr.py:577-753).

        ``fit_gpr``: True (restarts), 'simple' (one run from the current optimum), a dict of
        ``fit_gpr_hyperparameters`` arguments, or False (fixed theta: refactorise only).
        """
        fit_kwargs = None
        if fit_gpr is True:
            fit_kwargs = {}
        elif str(fit_gpr) == "simple":
            fit_kwargs = {"simple": True}
        elif isinstance(fit_gpr, Mapping):
            fit_kwargs = copy.deepcopy(dict(fit_gpr))
        elif fit_gpr is not False:
            raise ValueError("`fit_gpr` needs to be bool, 'simple', or a dict of args for the "
                             f"`fit_gpr_hyperparameters` method. Got {fit_gpr}.")
        do_fit = fit_kwargs is not None
        if do_fit:
            fit_classifier = True
        fit_preprocessors = bool(fit_classifier)
        force_fit = False
        if X is None and y is None:
            X, y = np.empty((0, self.d)), np.empty((0,))
            force_fit = do_fit
            if noise_level is not None:
                raise ValueError("Cannot give a noise level if X and y are not given.")
        elif X is None or y is None:
            raise ValueError("If passing X, y needs to be passed too, and viceversa.")
        X = np.asarray(X, dtype=float).reshape(-1, self.d) if np.size(X) else np.empty((0, self.d))
        y = np.atleast_1d(np.asarray(y, dtype=float))
        noise_valid = self._validate_noise_level(noise_level, len(y))
        self.n_last_appended = len(y)
        self.X_train_all = np.append(self.X_train_all, X, axis=0)
        self.y_train_all = np.append(self.y_train_all, y)
        self._update_noise_level(noise_valid)
        clf = self.infinities_classifier
        if clf is None:
            is_finite_all = np.full(len(self.y_train_all), True)
            X_fin, y_fin = np.copy(self.X_train_all), np.copy(self.y_train_all)
        else:
            thr_keep = self._diff_threshold_if_keep_n_finite(
                self.y_train_all, self.keep_min_finite, self._diff_threshold)
            is_finite_all = clf._is_finite_raw(self.y_train_all, thr_keep)
            X_fin = np.copy(self.X_train_all[is_finite_all])
            y_fin = np.copy(self.y_train_all[is_finite_all])
        if fit_preprocessors:
            self.preprocessing_X.fit(X_fin, y_fin)
            self.preprocessing_y.fit(X_fin, y_fin)
        self.X_train_all_ = self.preprocessing_X.transform(self.X_train_all)
        self.y_train_all_ = self.preprocessing_y.transform(self.y_train_all)
        noise_arr = (np.full(len(self.y_train_all_), self.noise_level)
                     if isinstance(self.noise_level, Number) else self.noise_level)
        self.noise_level_ = self.preprocessing_y.transform_scale(noise_arr)
        if clf is None:
            last_finite = np.full(self.n_last_appended, True)
        else:
            if fit_classifier:
                thr_ = self.preprocessing_y.transform_scale(thr_keep)
                pred = clf.fit(self.X_train_all_, self.y_train_all_, thr_)
                assert np.array_equal(is_finite_all, pred), \
                    "Infinities classifier miss-classified at least 1 point."
            last_finite = is_finite_all[len(is_finite_all) - self.n_last_appended:]
        self.n_last_appended_finite = int(sum(last_finite))
        if not self.n_last_appended_finite and not force_fit:
            return self
        # fixed theta, frozen pre-processors, a valid factor on the device: the factor can grow by border
        # rows instead of being rebuilt (the "lies" of the acquisition step)
        border = None
        if (not do_fit and not fit_preprocessors and self._dev_factor_ok and self._dev_train_ok
                and self.X_train_ is not None and self.kernel_ is not None and hasattr(self.device, "append_rows")):
            border = (self.X_train_, self.y_train_, np.asarray(self.alpha))
        self.X_train, self.y_train = X_fin, y_fin
        self.X_train_ = np.ascontiguousarray(self.preprocessing_X.transform(self.X_train))
        self.y_train_ = np.ascontiguousarray(self.preprocessing_y.transform(self.y_train))
        self.alpha = np.asarray(self.noise_level_)[is_finite_all] ** 2
        self.newly_appended_for_inv = self.n_last_appended_finite
        self._invalidate(train=True)
        if do_fit:
            self.fit_gpr_hyperparameters(**fit_kwargs)
        elif border is None or not self._border_update(*border):
            self._update_model()
        self.update_trust_region()
        return self

    def _border_update(self, X_old_, y_old_, alpha_old):
        """Extend the device factor by the rows appended last (``gpry_append_rows``) if the old
        training rows are an unchanged prefix of the new ones; False -> the caller refactorises."""
        n_old = len(y_old_)
        k = len(self.y_train_) - n_old
        if k <= 0 or not (np.array_equal(self.X_train_[:n_old], X_old_) and np.array_equal(self.y_train_[:n_old], y_old_)
                          and np.array_equal(self.alpha[:n_old], np.broadcast_to(alpha_old, (n_old,)))):
            return False
        info = self.device.append_rows(self.X_train_[n_old:], self.y_train_[n_old:], self.alpha[n_old:])
        if info != 0:
            return False        # the full path raises the reference's "not positive definite" error
        self._dev_train_ok = self._dev_factor_ok = True
        self._factor_epoch = getattr(self, "_factor_epoch", 0) + 1
        self.n_border_updates = getattr(self, "n_border_updates", 0) + 1
        self.newly_appended_for_inv = 0
        return True

    def fit(self, X, y):
        """Thin alias (the reference never calls sklearn's ``fit``; SURVEY.md preamble)."""
        return self.append_to_data(X, y, fit_gpr=True)

    def _validate_noise_level(self, noise_level, n_train):
        """Noise of a batch of ``n_train`` new points as it will be stored: an array once any array was given
        (now or earlier), a scalar or None otherwise (gpry/gpr.py:755-785)."""
        per_point_so_far = np.iterable(self.noise_level)
        if noise_level is None:
            if per_point_so_far:
                raise ValueError("Need to pass non-null noise_level (scalar or array) because "
                                 "concrete values were given earlier for the training points.")
            return None
        if n_train == 0:
            raise ValueError("noise_level must be None if not fitting to new points.")
        if np.iterable(noise_level):
            levels = np.atleast_1d(noise_level)
            if len(levels) != n_train:
                raise ValueError("noise_level must be an array with same number of entries as y, "
                                 f"but len(n)={len(levels)} != len(y)={n_train})")
            return levels
        if not isinstance(noise_level, Number):
            raise ValueError("noise_level needs to be an iterable, number or None. "
                             f"Got type(noise_level)={type(noise_level)}")
        return np.full(n_train, noise_level) if per_point_so_far else noise_level

    def _update_noise_level(self, noise_level):
        """Merge the (validated) noise of the points just appended into ``self.noise_level`` (gpry/gpr.py:787-817)."""
        chatty = self.verbose > 1
        if np.iterable(noise_level):
            known = self.noise_level
            if not np.iterable(known):          # a scalar so far: spell it out for the points already in
                if chatty:
                    warnings.warn("A new noise level has been assigned to the updated training set "
                                  "while the old training set has a single scalar noise level: "
                                  f"{known}. Converting to individual levels!")
                known = np.full(len(self.y_train_all) - len(noise_level), known)
            self.noise_level = np.concatenate((known, noise_level))
        elif isinstance(noise_level, Number):
            assert not np.iterable(self.noise_level)
            if not np.isclose(noise_level, self.noise_level):
                if chatty:
                    warnings.warn("Overwriting the noise level with a scalar. Make sure that "
                                  "kernel's hyperparamters are refitted.")
                self.noise_level = noise_level

    def remove_from_data(self, position, fit=True):
        raise NotImplementedError("This function is outdated and needs review.")

    # ---- marginal likelihood and fit ------------------------------------------------------
    def log_marginal_likelihood(self, theta=None, eval_gradient=False, clone_kernel=True):
        """Device evaluation of sklearn:_gpr.py:574-652 (counted as gpry/gpr.py:876-881)."""
        self.n_eval_loglike += 1
        if theta is None:
            if eval_gradient:
                raise ValueError("Gradient can only be evaluated for theta!=None")
            return self.log_marginal_likelihood_value_
        theta = np.asarray(theta, dtype=float)
        kernel = self.kernel_ if self.kernel_ is not None else clone(self.kernel)
        fast = hasattr(kernel, "set_theta_and_full")
        if clone_kernel:
            kernel = kernel.clone_with_theta(theta)
            kid, theta_full = kernel.device_spec(self.d)
        else:
            if self.kernel_ is None:
                self.kernel_ = kernel
            if fast:                      # same side effect as the reference: kernel_.theta = theta
                kid, theta_full = kernel.set_theta_and_full(theta, self.d)
            else:
                kernel.theta = theta
                kid, theta_full = kernel.device_spec(self.d)
            self._dev_factor_ok = False
            self._host_factor = {}
            self._kb = None
        if not self._dev_train_ok:
            self._upload_train()
        dev = self._dev if self._dev is not None else self.device
        if dev.N != len(self.y_train_):
            raise RuntimeError("device training set out of sync")
        if eval_gradient and getattr(self, "_seq_fit_tp", False):
            # a sequential fit under the throughput schedule (the comparator of the side-by-side fit, fit_gpr_hyperparameters):
            # the chain of the many for ONE theta -- the bits a theta has in any batch of that schedule
            lb, gb, _ = dev.lml_batch(np.asarray(theta_full, dtype=float)[None, :], True)
            lml, grad_full = float(lb[0]), gb[0]
            if not np.isfinite(lml):
                return -np.inf, np.zeros_like(theta)
            return lml, (kernel.grad_from_full_fast(grad_full, self.d) if fast else kernel.grad_from_full(grad_full, self.d))
        if eval_gradient:
            lml, grad_full, _ = dev.lml(theta_full, True)
            if not np.isfinite(lml):
                return -np.inf, np.zeros_like(theta)
            return lml, (kernel.grad_from_full_fast(grad_full, self.d) if fast else kernel.grad_from_full(grad_full, self.d))
        lml, _ = dev.lml(theta_full, False)
        return lml if np.isfinite(lml) else -np.inf

    def fit_gpr_hyperparameters(self, simple=False, start_from_current=True, n_restarts=None,
                                hyperparameter_bounds=None):
        """Maximise the marginal likelihood over theta (gpry/gpr.py:883-994): same restart
        schedule, RNG order (:969-978) and optimiser (scipy L-BFGS-B with jac) as the
        reference; the objective and its gradient are evaluated on the device."""
        if simple:
            start_from_current, n_restarts = True, 1
        if not self._fitted:
            start_from_current = False
        if n_restarts is None:
            n_restarts = self.n_restarts_optimizer
        if self.kernel_ is None:
            self.kernel_ = clone(self.kernel)
        reasons = []
        if self.optimizer is None:
            reasons.append("no optimizer has been specified")
        if self.kernel.n_dims == 0:
            reasons.append("the kernel has no hyperparamenters")
        if n_restarts <= 0:
            reasons.append("the number of optimizer restarts requested is 0.")
        if reasons:
            warnings.warn(f"Hyper-parameters not (re)fit. Reason(s): {'; '.join(reasons)}.")
            self.log_marginal_likelihood_value_ = self.log_marginal_likelihood(
                self.kernel_.theta, clone_kernel=False)
            self._update_model()
            return self

        def obj_func(theta, eval_gradient=True):
            if eval_gradient:
                lml, grad = self.log_marginal_likelihood(theta, eval_gradient=True,
                                                         clone_kernel=False)
                return -lml, -grad
            return -self.log_marginal_likelihood(theta, clone_kernel=False)

        if hyperparameter_bounds is None:
            hyperparameter_bounds = self.kernel_.bounds
        if n_restarts - int(start_from_current):
            if not np.isfinite(hyperparameter_bounds).all():
                raise ValueError(
                    "There is at least one optimizer run the requires sampling from the "
                    "hyperparameters' prior, but it has not finite density, because not all "
                    "bounds are finite. You can pass some finite bounds manually using "
                    "``hyperparameter_bounds``.")
        self._rng = check_random_state(self.random_state)
        ctx_devs = (fit_context_devices(getattr(self.device, "device", 0), n_restarts, getattr(self, "fit_devices", None))
                    if self.optimizer == "fmin_l_bfgs_b" else [0])
        self._side_by_side_why = ""
        side_by_side = n_restarts > 1 and self._can_step_restarts_together()
        if side_by_side or len(ctx_devs) > 1:
            # the optimiser never touches the RNG: drawing the start points up front gives the reference's
            # sequence (gpry/gpr.py:969-978)
            starts = [np.array(self.kernel_.theta) if (it == 0 and start_from_current) else
                      self._rng.uniform(hyperparameter_bounds[:, 0], hyperparameter_bounds[:, 1])
                      for it in range(n_restarts)]
            optima = (self._restarts_side_by_side(starts, hyperparameter_bounds) if side_by_side
                      else self._concurrent_restarts(starts, hyperparameter_bounds, ctx_devs))
        else:
            # The sequential loop evaluates with single gpry_lml calls (the latency schedule).  ``GPRY_HIP_FIT_SCHEDULE=throughput``
            # makes it the comparator of a throughput-schedule fit instead: every evaluation through the chain of the many with
            # one theta -- the side-by-side fit under that schedule then equals this loop bit for bit
            seq_tp = (n_restarts > 1 and os.environ.get("GPRY_HIP_FIT_SCHEDULE", "").lower() == "throughput"
                      and hasattr(self.device, "set_option") and 128 < self.n <= int(getattr(self.device, "lml_batch_max", 128)))
            optima = []
            try:
                if seq_tp:
                    self._upload_train()
                    self.device.set_option("lml_schedule", 1)
                    self._seq_fit_tp = True
                for iteration in range(n_restarts):
                    if iteration == 0 and start_from_current:
                        theta0 = self.kernel_.theta
                    else:
                        theta0 = self._rng.uniform(hyperparameter_bounds[:, 0], hyperparameter_bounds[:, 1])
                    optima.append(self._constrained_optimization(obj_func, theta0, hyperparameter_bounds))
            finally:
                if seq_tp:
                    self._seq_fit_tp = False
                    self.device.set_option("lml_schedule", 0)
            self.fit_stats = {"contexts": 1, "devices": [getattr(self.device, "device", 0)], "evals_per_context": None,
                              "schedule": "throughput" if seq_tp else "latency"}
        # how the restarts ran, for whoever wonders why a fit was slow: side by side (one batched objective per round) or
        # one after another / farmed over contexts, and -- when scipy's private routine was the obstacle -- why
        self.fit_stats = dict(getattr(self, "fit_stats", None) or {})
        self.fit_stats["side_by_side"] = bool(side_by_side)
        self.fit_stats["why"] = "" if side_by_side else (self._side_by_side_why or
                                                         ("one restart" if n_restarts <= 1 else "not applicable to this model / optimizer"))
        values = [o[1] for o in optima]
        self.log_marginal_likelihood_value_ = -np.min(values)
        self.kernel_.theta = optima[int(np.argmin(values))][0]
        self._invalidate()
        self._update_model()
        self._fitted = True
        return self

    def _can_step_restarts_together(self):
        """Can ``gpry_lml_batch`` evaluate a round's thetas together?  N <= 128, d <= 16: ONE launch with a workgroup per
        run (the single-launch objective); above that and up to the device's ``lml_batch`` limit (4096): ONE chain of
        launches whose every kernel carries all runs.  Then the optimiser runs of a fit advance side by side."""
        if (self.optimizer != "fmin_l_bfgs_b" or getattr(self, "fit_lockstep", True) is False
                or os.environ.get("GPRY_HIP_FIT_LOCKSTEP", "1") == "0"):
            return False
        if not hasattr(self.device, "lml_batch"):
            return False
        if self.n <= 128:
            if self.d > 16:
                return False
        elif self.n > int(getattr(self.device, "lml_batch_max", 128)):
            return False
        from gpry_amd import lockstep
        if lockstep.available():
            return True
        # everything else allows the side-by-side form and scipy's routine does not: say so once (the fit below is 3-7x
        # slower) and leave the reason where a caller can find it (fit_stats["why"])
        lockstep.warn_once("multi-restart hyper-parameter fits")
        self._side_by_side_why = lockstep.why()
        return False

    def _restarts_side_by_side(self, starts, bounds):
        try:
            return self._restarts_side_by_side_impl(starts, bounds)
        finally:
            # the model's own context goes back to the schedule of the single evaluations, whatever happened
            if getattr(self, "_fit_sched", "latency") == "throughput" and hasattr(self.device, "set_option"):
                try:
                    self.device.set_option("lml_schedule", 0)
                except Exception:
                    pass
            self._fit_sched = "latency"

    def _restarts_side_by_side_impl(self, starts, bounds):
        """The runs of a multi-restart fit stepped together (``gpry_amd.lockstep``: scipy's own L-BFGS-B routine, one
        reverse-communication call per run and round), the objective of a round evaluated for all runs in one
        ``gpry_lml_batch``.  Every run sees the values -- to the bit -- and takes the steps it would take alone, so
        the optima and the selected one are those of the sequential loop.

        Above a few hundred points the runs are dealt out (run i to group i mod k) over k INDEPENDENT groups, each with a
        lock-step driver, a host thread and a device context of its own (``batch_contexts``; the contexts follow
        ``fit_context_devices``, i.e. all GPUs of a single-process run): a round costs the latency of the kernel chain
        whatever its width, and the chains of the groups run beside each other on the GPU -- the thread farm's overlap
        with the batching inside every thread.  No group waits for another; a run never changes group."""
        from gpry_amd import lockstep
        self._upload_train()
        dev, d = self.device, self.d
        n_runs = len(starts)
        devs = [dev]
        k = 1
        # which schedule of the batched objective (fit_schedule): set on the model's own context here, copied to the others below
        sched, tp_groups = ("latency", None)
        if hasattr(dev, "set_option") and 128 < self.n <= int(getattr(dev, "lml_batch_max", 128)):
            sched, tp_groups = fit_schedule(self.n, n_runs)
            try:
                dev.set_option("lml_schedule", 1 if sched == "throughput" else 0)
            except Exception:       # a build without the option
                sched = "latency"
        self._fit_sched = sched
        n_groups = batch_contexts(self.n) if sched == "latency" else tp_groups
        if hasattr(dev, "set_train") and n_groups > 1:
            # one GPU: batch_contexts() groups on it; several GPUs in the process: the groups go where fit_context_devices
            # deals the contexts of a fit (round-robin over the GPUs, its first entries on distinct ones)
            ctx_devs = fit_context_devices(getattr(dev, "device", 0), n_runs, getattr(self, "fit_devices", None))
            k_max = len(ctx_devs) if len(set(ctx_devs)) > 1 else n_groups
            k = min(k_max, max(1, n_runs // _BATCH_GROUP_MIN_RUNS))
        if k > 1:
            want = [ctx_devs[i] if i < len(ctx_devs) else ctx_devs[0] for i in range(1, k)]
            kid, theta_full0 = self.kernel_.device_spec(self.d)
            spare = list(self._fit_devs)
            for idx in want:
                hit = next((pr for pr in spare if pr[0] == idx), None)
                if hit is None:
                    hit = (idx, type(dev)(idx))
                    self._fit_devs.append(hit)
                else:
                    spare.remove(hit)
                copy_fit_options(dev, hit[1])
                hit[1].set_train(self.X_train_, self.y_train_, self.alpha)
                hit[1].set_theta(kid, theta_full0)
                devs.append(hit[1])
        starts = np.array(starts, dtype=float)
        bounds = np.asarray(bounds, dtype=float)
        counts = [0] * k

        def make_fg(g):
            kern = clone(self.kernel_)              # (kernel objects are not shared between threads)
            fast = hasattr(kern, "set_theta_and_full")
            dv = devs[g]
            # theta = [log C, log l_1 .. log l_d] (the automatic kernel): the device vector is log(exp(theta)) row by row --
            # what ``device_spec`` returns after ``kernel.theta = theta`` -- and the gradient needs no chain rule.  Going
            # through the kernel object cost 15 us per theta, more than the device's share of a round below ~300 points.
            plain = False
            if fast:
                _, _, n_c, n_l, _, const_first = kern._layout()
                plain = n_c == 1 and n_l == d and const_first

            def fg(Thetas):
                if plain:
                    T = np.asarray(Thetas, dtype=float)
                    fulls = np.empty_like(T)
                    for j in range(len(T)):
                        fulls[j] = np.log(np.exp(T[j]))       # (per row, as one evaluation does it: same bits)
                    lml, grad_full, _ = dv.lml_batch(fulls, True)
                    counts[g] += len(T)
                    bad = ~np.isfinite(lml)
                    F, G = -lml, -grad_full[:, :d + 1]
                    if bad.any():
                        F[bad] = np.inf
                        G[bad] = 0.0
                    return F, G
                fulls = []
                for th in Thetas:
                    if fast:
                        fulls.append(np.array(kern.set_theta_and_full(th, d)[1], dtype=float))
                    else:
                        kern.theta = np.asarray(th, dtype=float)
                        fulls.append(np.array(kern.device_spec(d)[1], dtype=float))
                lml, grad_full, _ = dv.lml_batch(np.array(fulls), True)
                counts[g] += len(Thetas)
                F, G = np.empty(len(Thetas)), np.zeros((len(Thetas), len(Thetas[0])))
                for j, th in enumerate(Thetas):
                    if not np.isfinite(lml[j]):
                        F[j] = np.inf
                        continue
                    if fast:
                        kern.set_theta_and_full(th, d)
                        G[j] = -kern.grad_from_full_fast(grad_full[j], d)
                    else:
                        kern.theta = np.asarray(th, dtype=float)
                        G[j] = -kern.grad_from_full(grad_full[j], d)
                    F[j] = -lml[j]
                return F, G
            return fg

        members = [list(range(g, n_runs, k)) for g in range(k)]
        if k == 1:
            out = [lockstep.minimize_lockstep(make_fg(0), starts, bounds)]
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=k) as pool:
                futs = [pool.submit(lockstep.minimize_lockstep, make_fg(g), starts[members[g]], bounds) for g in range(k)]
                out = [f.result() for f in futs]
        X, F, nfev = np.empty_like(starts), np.empty(n_runs), np.zeros(n_runs, dtype=int)
        for g in range(k):
            X[members[g]], F[members[g]], nfev[members[g]] = out[g]
        self.n_eval_loglike += sum(counts)
        self.fit_stats = {"contexts": k, "devices": [getattr(dv, "device", 0) for dv in devs], "side_by_side": True,
                          "evals_per_run": [int(v) for v in nfev], "evals_per_context": list(counts), "schedule": sched}
        try:        # a fit that ran short of device memory halves its batches (and is slower for it): say so
            self.fit_stats["batch_shrinks"] = int(sum(dv.timing("lml_batch_shrinks")[1] for dv in devs if hasattr(dv, "timing")))
        except Exception:
            pass
        return [(X[i], F[i]) for i in range(n_runs)]

    def _concurrent_restarts(self, starts, bounds, ctx_devs):
        """The optimiser runs of a multi-restart fit are independent: they are worked off by one host
        thread per entry of ``ctx_devs`` (device indices, entry 0 = the model's own context), each with its
        own device context holding the same training set (a context is not re-entrant, distinct contexts
        may be driven from distinct threads; ctypes releases the GIL while an evaluation runs).  On one
        GPU, one LML evaluation leaves most of it idle in its latency-bound panel steps: 2 / 3 contexts
        give 1.6x / 1.9x the evaluations per second at N=4096 and 1.3x / 1.4x at N=8192
        (``tools/ab_concurrent_lml.py``); with several GPUs in the process the contexts are spread over
        them (``fit_context_devices``: the restart farm of BASELINE configs[4] in ONE process).  Every
        run is deterministic and evaluated exactly as in the sequential loop, so the optima -- and the
        selected one -- do not depend on the schedule or on which GPU ran them."""
        import queue
        import threading
        self._upload_train()
        kern0 = self.kernel_
        kid, theta_full0 = kern0.device_spec(self.d)
        n_ctx = len(ctx_devs)
        devs = [self.device]
        spare = list(self._fit_devs)
        used = []
        for idx in ctx_devs[1:]:
            hit = next((pr for pr in spare if pr[0] == idx), None)
            if hit is None:
                hit = (idx, type(self.device)(idx))         # same kind as the main one
                self._fit_devs.append(hit)
            else:
                spare.remove(hit)
            used.append(hit)

        def _replicate(dv):
            copy_fit_options(self.device, dv)
            dv.set_train(self.X_train_, self.y_train_, self.alpha)
            dv.set_theta(kid, theta_full0)

        if len({idx for idx, _ in used}) > 1:       # several GPUs: their uploads run side by side
            ths = [threading.Thread(target=_replicate, args=(dv,)) for _, dv in used]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        else:
            for _, dv in used:
                _replicate(dv)
        devs += [dv for _, dv in used]
        self.fit_stats = {"contexts": n_ctx, "devices": list(ctx_devs), "evals_per_context": None}
        todo = queue.Queue()
        for it, th in enumerate(starts):
            todo.put((it, th))
        optima = [None] * len(starts)
        counts = [0] * n_ctx
        errors = []

        def worker(k):
            dev, kern = devs[k], clone(kern0)

            fast = hasattr(kern, "set_theta_and_full")

            def obj_func(theta, eval_gradient=True):
                counts[k] += 1
                if fast:
                    full = kern.set_theta_and_full(theta, self.d)[1]
                else:
                    kern.theta = np.asarray(theta, dtype=float)
                    full = kern.device_spec(self.d)[1]
                lml, grad_full, _ = dev.lml(full, True)
                if not np.isfinite(lml):
                    return np.inf, np.zeros_like(theta)
                return -lml, -(kern.grad_from_full_fast(grad_full, self.d) if fast else kern.grad_from_full(grad_full, self.d))

            try:
                while True:
                    try:
                        it, th = todo.get_nowait()
                    except queue.Empty:
                        return
                    optima[it] = self._constrained_optimization(obj_func, th, bounds)
            except BaseException as e:      # re-raised on the calling thread
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_ctx)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        self.n_eval_loglike += sum(counts)
        self.fit_stats["evals_per_context"] = list(counts)
        if errors:
            raise errors[0]
        return optima

    def _constrained_optimization(self, obj_func, initial_theta, bounds):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if self.optimizer == "fmin_l_bfgs_b":
                res = scipy.optimize.minimize(obj_func, initial_theta, method="L-BFGS-B", jac=True,
                                              bounds=bounds)
                return res.x, res.fun
            if callable(self.optimizer):
                return self.optimizer(obj_func, initial_theta, bounds=bounds)
            raise ValueError("Unknown optimizer %s." % self.optimizer)

    def _update_model(self):
        """Refactorise at fixed theta (gpry/gpr.py:996-1020)."""
        if self.newly_appended_for_inv < 1:
            warnings.warn("No new points have been appended to the model.")
            return self
        if self.kernel_ is None:
            self.kernel_ = clone(self.kernel)
        self._invalidate()
        self._ensure_factor()
        self.newly_appended_for_inv = 0
        return self

    # ---- prediction -----------------------------------------------------------------------
    def _masks(self, X, validate, ignore_trust_region):
        """Host-side gates turned into the device's per-candidate mask bits."""
        mask = None
        if self.trust_bounds is not None and not ignore_trust_region:
            outside = np.logical_not(is_in_bounds(X, self.trust_bounds, check_shape=False))
            mask = outside.astype(np.uint8) * _lib.MASK_OUTSIDE_TRUST
        if self.infinities_classifier is not None:
            X_ = self.preprocessing_X.transform(X)
            finite = self.infinities_classifier.predict(np.ascontiguousarray(X_), validate=validate)
            bits = np.logical_not(finite).astype(np.uint8) * _lib.MASK_CLASSIFIED_INF
            mask = bits if mask is None else (mask | bits)
        return mask

    def _push_gates(self, ignore_trust_region=False, sinks=None):
        """Hand the classifier's decision function and the trust box to the device so that the
        sweep computes the per-candidate mask itself.  Returns True
        if the device now covers every gate ``_masks`` would apply, False if the caller must
        keep using ``_masks`` (classifier without a device form); the device gates are switched
        off in that case.  ``sinks``: what receives the gates (default: this model's context; a
        sharded sweep passes its device group)."""
        clf = self.infinities_classifier
        sinks = [self.device] if sinks is None else list(sinks)
        trust = None if (self.trust_bounds is None or ignore_trust_region) else self.trust_bounds
        params = None
        ok = True
        if clf is not None:
            always_finite = getattr(clf, "all_finite", False) and getattr(clf, "y_train", None) is not None
            params = clf.device_params() if hasattr(clf, "device_params") else None
            if params is None and not always_finite:
                for snk in sinks:
                    snk.set_gates()
                ok = False
        for snk in (sinks if ok else ()):
            if params is None:
                snk.set_gates(trust_bounds=trust)
            else:
                sv, coef, gamma, intercept, pos = params
                snk.set_gates(sv, coef, gamma, intercept, pos, trust_bounds=trust)
        if any(snk is self._dev for snk in sinks):
            self._dev_gates = (self._gates_key(ignore_trust_region), ok)
        return ok

    def _gates_key(self, ignore_trust_region):
        clf = self.infinities_classifier
        trust = None if (self.trust_bounds is None or ignore_trust_region) else np.asarray(self.trust_bounds).tobytes()
        return (id(clf), getattr(clf, "_fit_count", 0), trust, id(self._dev))

    def _sync_gates(self, ignore_trust_region=False):
        """``_push_gates`` for the point-by-point callers: the gates travel to the device only when classifier,
        trust box or context changed since the last push (an upload per call would cost more than the call)."""
        held = self._dev_gates
        if held is not None and held[0] == self._gates_key(ignore_trust_region):
            return held[1]
        return self._push_gates(ignore_trust_region)

    def _validate_X(self, X, validate):
        if validate:
            X = np.asarray(X, dtype=float)
            if X.ndim != 2:
                raise ValueError(f"Expected 2D array, got array with shape {X.shape}")
            if not np.all(np.isfinite(X)):
                raise ValueError("Input X contains NaN or infinity.")
        return X

    def predict(self, X, return_std=False, return_cov=False, return_mean_grad=False,
                return_std_grad=False, validate=True, ignore_trust_region=False):
        """Posterior mean (and std) at ``X`` in untransformed units (gpry/gpr.py:1022-1273).

        Clipping, classifier and trust-region gates behave as in the reference.  The x-gradient
        outputs (one point at a time, :1236-1266) are taken with respect to the TRANSFORMED
        coordinates, scaled once (mean) and twice (std) by ``std_y`` -- as the reference does.
        """
        self.n_eval += len(X)
        if return_std_grad and not (return_std and return_mean_grad):
            raise ValueError("Not returning std_gradient without returning the std and the mean grad.")
        if X.shape[0] != 1 and (return_mean_grad or return_std_grad):
            raise ValueError("Mean grad and std grad not implemented for n_samples > 1")
        X = self._validate_X(X, validate)
        if self.X_train_ is None:  # not fit: GP prior
            y_mean = np.zeros(X.shape[0])
            if self.trust_bounds is not None and not ignore_trust_region:
                y_mean[~is_in_bounds(X, self.trust_bounds)] = self.minus_inf_value
            out = [y_mean]
            if return_std:
                out.append(np.sqrt(self.kernel.diag(X)))
            if return_mean_grad:
                out.append(np.zeros_like(X))
                if return_std_grad:
                    out.append(np.zeros_like(X))
            return tuple(out) if len(out) > 1 else out[0]
        self._ensure_factor()
        self._push_affine()
        # classifier and trust box: on the device with the points (``gpry_predict`` with "predict_gates") when they have
        # a device form; the x-gradient branch below needs the verdict on the host
        # (the x-gradient branch reads the verdict: the one-point call returns it, the older entry points do not)
        one_point = return_mean_grad and return_std and hasattr(self.device, "predict_point")
        if ((not return_mean_grad or one_point) and getattr(self.device, "applies_gates_in_predict", False)
                and (self.infinities_classifier is not None or self.trust_bounds is not None)
                and self._sync_gates(ignore_trust_region)):
            mask = None
        else:
            if self._dev_gates is not None and self._dev_gates[1] and getattr(self.device, "applies_gates_in_predict", False):
                self.device.set_gates()          # host verdicts for this call: nothing may be ORed in on the device
                self._dev_gates = None
            mask = self._masks(X, validate, ignore_trust_region)
        # one point with std and gradients (an acquisition optimiser's step): a single device call where the device has one
        point_call = one_point and not (mask is not None and bool(mask[0] & _lib.MASK_CLASSIFIED_INF))
        verdict = 0 if mask is None else int(mask[0])
        if point_call:
            m1, s1, mg, kg, verdict = self.device.predict_point(X[0], mask_bits=verdict, want_kinv=bool(return_std_grad))
            y_mean, y_std = np.array([m1]), np.array([s1])
        else:
            res = self.device.predict(X, return_std=return_std, mask=mask)
            y_mean = res[0] if return_std else res
            y_std = res[1] if return_std else None
        if self.minus_inf_value != -np.inf:
            y_mean[np.isneginf(y_mean)] = self.minus_inf_value
        if not return_mean_grad:
            return (y_mean, y_std) if return_std else y_mean
        # ---- one point: gradients (gpry/gpr.py:1236-1266)
        n_dims = X.shape[1]
        classified_inf = bool(verdict & _lib.MASK_CLASSIFIED_INF)
        if classified_inf:     # :1157-1171: mean -inf, std 0, mean gradient +inf, std gradient 0
            grad_mean = np.ones((1, n_dims)) * self.inf_value
            grad_std = np.zeros((1, n_dims))
        else:
            _, std_y = self._y_affine()
            # (np.allclose(y_std, 0) of the reference for the single value there is: |std| <= 1e-8, False for NaN)
            want_std_grad = bool(return_std_grad) and not abs(float(y_std[0])) <= 1e-8
            if not point_call:
                mg, kg = self.device.predict_grad(X[0], want_kinv=want_std_grad)
            grad_mean = mg * std_y
            grad_std = np.zeros(n_dims)
            if want_std_grad:
                y_std_untransformed = y_std / std_y
                grad_std = -kg / y_std_untransformed * std_y * std_y
            if self.infinities_classifier is not None:   # the reference re-embeds into (n, d) arrays
                grad_mean, grad_std = grad_mean[None, :], grad_std[None, :]
        if return_std_grad:
            return y_mean, y_std, grad_mean, grad_std
        return (y_mean, y_std, grad_mean) if return_std else (y_mean, grad_mean)

    def predict_with_gradients(self, X, validate=True, ignore_trust_region=False):
        """``predict(x, return_std=True, return_mean_grad=True, return_std_grad=True)`` for every row of
        ``X`` in ONE device call (``gpry_predict_grad_batch``): ``(mean (m,), std (m,), mean_grad (m, d),
        std_grad (m, d))`` with the reference's single-point conventions row by row (gpry/gpr.py:1236-1266:
        gradients in the transformed coordinates, scaled once / twice by std_y; classifier-rejected
        points give -inf, 0, +inf, 0; a vanishing std gives a zero std gradient).  The reference
        evaluates one point per call ("not implemented for n_samples > 1"); a batch of acquisition
        optimiser restarts evaluated side by side reads V once instead of twice per point."""
        X = self._validate_X(np.atleast_2d(X), validate)
        m, n_dims = X.shape
        self.n_eval += m
        if self.X_train_ is None:
            raise ValueError("predict_with_gradients needs a model with training data")
        self._ensure_factor()
        self._push_affine()
        mask = self._masks(X, validate, ignore_trust_region)
        mean, std, mg, kg = self.device.predict_grad_batch(X, want_kinv=True)
        _, std_y = self._y_affine()
        grad_mean = mg * std_y
        with np.errstate(divide="ignore", invalid="ignore"):
            grad_std = -kg / (std / std_y)[:, None] * std_y * std_y
        grad_std[np.isclose(std, 0.0)] = 0.0            # gpry/gpr.py:1253-1254
        if mask is not None:
            mean[mask != 0] = self.minus_inf_value
            rejected = (mask & _lib.MASK_CLASSIFIED_INF) != 0
            std[rejected] = 0.0
            grad_mean[rejected] = self.inf_value        # gpry/gpr.py:1157-1171
            grad_std[rejected] = 0.0
        return mean, std, grad_mean, grad_std

    def predict_std(self, X, validate=True):
        """gpry/gpr.py:1275-1352 (no trust-region gate, classifier-masked rows give 0)."""
        self.n_eval += len(X)
        X = self._validate_X(X, validate)
        if self.X_train_ is None:
            return np.sqrt(self.kernel.diag(X))
        self._ensure_factor()
        self._push_affine()
        mask = None
        on_device = getattr(self.device, "applies_gates_in_predict", False)
        if on_device and self.infinities_classifier is not None and self._sync_gates(ignore_trust_region=True):
            pass        # the CURRENT classifier sits on the device (re-pushed if it was refitted since the last push)
        else:
            # gpry_predict ORs whatever gates the context holds into the mask: a classifier that was refitted since the
            # last push (append_to_data) must not vote on this call
            if on_device and self._dev_gates is not None and self._dev_gates[1]:
                self.device.set_gates()
                self._dev_gates = None
            if self.infinities_classifier is not None:
                X_ = self.preprocessing_X.transform(X)
                finite = self.infinities_classifier.predict(np.ascontiguousarray(X_), validate=validate)
                mask = np.logical_not(finite).astype(np.uint8) * _lib.MASK_CLASSIFIED_INF
        return self.device.predict(X, return_std=True, mask=mask)[1]

    # ---- Kriging-believer conditioning (used by RankedPool.cache_model) ---------------------
    def kb_session(self):
        from gpry_amd.kriging import KBSession
        self._ensure_factor()
        if self._kb is None:
            self._kb = KBSession(self)
        return self._kb

    def conditioned(self, X, y):
        """Model augmented by ``(X, y)`` at fixed theta and frozen pre-processors -- what
        ``deepcopy(gpr).append_to_data(X, y, fit_gpr=False, fit_classifier=False)`` yields in
        the reference (gpry/gp_acquisition.py:1550-1553) -- as a bordered factor."""
        return self.kb_session().conditioned(np.atleast_2d(X), np.atleast_1d(y))

    # ---- copies and pickles ------------------------------------------------------------------
    def __deepcopy__(self, memo):
        """Same field semantics as gpry/gpr.py:1354-1433 (a fresh object from the constructor
        arguments that the reference forwards, then the data fields); device state is not
        shared: the copy rebuilds its factor on first use."""
        c = GaussianProcessRegressor(
            kernel=copy.deepcopy(self.kernel), noise_level=copy.deepcopy(self.noise_level),
            optimizer=self.optimizer, n_restarts_optimizer=self.n_restarts_optimizer,
            preprocessing_X=self.preprocessing_X, preprocessing_y=self.preprocessing_y,
            bounds=self.bounds, random_state=self.random_state, account_for_inf=None)
        for name in ("n_eval", "n_eval_loglike", "noise_level_", "alpha", "n_last_appended",
                     "n_last_appended_finite", "newly_appended_for_inv", "_fitted",
                     "log_marginal_likelihood_value_"):
            if hasattr(self, name):
                setattr(c, name, getattr(self, name))
        for name in ("X_train", "y_train", "X_train_", "y_train_", "X_train_all", "y_train_all",
                     "X_train_all_", "y_train_all_"):
            v = getattr(self, name, None)
            setattr(c, name, None if v is None else np.copy(v))
        for name in ("kernel_", "infinities_classifier", "_diff_threshold", "keep_min_finite",
                     "trust_region_factor", "trust_region_nstd", "inf_value", "minus_inf_value"):
            if hasattr(self, name):
                setattr(c, name, copy.deepcopy(getattr(self, name)))
        if self._dev is not None:
            c._device_kind = type(self._dev)       # a copy gets a context of the same kind (tests: a double)
        return c

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in ("_dev", "_kb", "_host_factor", "_fit_devs", "_affine_cache", "_dev_affine", "_dev_affine_on", "_dev_gates"):
            state.pop(k, None)
        state["_dev_train_ok"] = False
        state["_dev_factor_ok"] = False
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._init_device_state()

    # ---- helpers shared with the reference ------------------------------------------------------
    @staticmethod
    def compute_threshold_given_sigma(n_sigma, n_dimensions):
        return delta_logp_of_1d_nstd(n_sigma, n_dimensions)

    @staticmethod
    def _diff_threshold_if_keep_n_finite(y, n, reference_diff_threshold, epsilon=1e-6):
        """gpry/gpr.py:1476-1488."""
        if n is None or n <= 1:
            return reference_diff_threshold
        ys = np.sort(y)
        return max(reference_diff_threshold, ys[-1] - ys[-min(n, len(ys))] + epsilon)
