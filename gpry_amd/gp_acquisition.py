"""NORA batch acquisition on the device sweep, with GPry's interface.

Host mirror of ``gpry/gp_acquisition.py``: ``GenericGPAcquisition`` (:38), ``NORA``
(:525, ``multi_add`` :971-1108) and ``RankedPool`` (:1194-1670).  What changes is where
the work happens:

* the evaluation of (mean, std, acquisition) on the whole candidate pool -- in the
  reference ``mpi.compute_y_parallel`` -> ``gpr.predict`` (gpry/mpi.py:182-218) and
  ``LogExp.f`` on host arrays -- is one fused device sweep (``gpry_sweep_logexp``);
* the descending-acquisition stream that ``RankedPool.add`` walks (:1328-1333) is
  produced by an exact device top-k (``gpry_sweep_topk``); the walk stops at the
  reference's own early-out (:1432), and the shortlist is extended until every
  candidate outside it provably hits that early-out, so the pool equals the one the
  reference computes from the full sorted stream;
* conditioned models come from ``gpry_amd.kriging`` (bordered factor) instead of
  ``deepcopy`` + refit;
* with several ranks (one process per GPU) each rank sweeps a contiguous shard and the
  per-rank shortlists are merged after one all-gather (RCCL), replacing the gathers of
  per-rank pools (:1148-1191);
* in ONE process (what ``gpry.Runner`` is without mpi4py, gpry/mpi.py:18-28) the same sharding
  runs over a device group (``devices=``, ``gpry_group_*``): every visible GPU by default.
"""
import os
import inspect
import sys
import warnings
from collections.abc import Mapping
from copy import deepcopy
from functools import partial
from time import time

import numpy as np

import scipy.optimize

import gpry_amd.acquisition_functions as gpryacqfuncs
from gpry_amd.proposal import CentroidsProposer, PartialProposer, Proposer
from gpry_amd.tools import get_Xnumber, get_random_generator, is_in_bounds, remove_0_weight_samples


def builtin_names():
    return [name for name, obj in inspect.getmembers(sys.modules[__name__])
            if inspect.isclass(obj) and issubclass(obj, GenericGPAcquisition)
            and obj is not GenericGPAcquisition]


class NestedSamplerNotInstalledError(Exception):
    """The requested external nested sampler is not importable."""


class GenericGPAcquisition:
    """Acquisition engine base: resolves the acquisition function (gp_acquisition.py:41-82)."""

    def __init__(self, bounds, preprocessing_X=None, verbose=1, acq_func="LogExp"):
        self.bounds_ = np.array(bounds).copy()
        self.n_d = self.bounds_.shape[0]
        self.preprocessing_X = preprocessing_X
        self.verbose = verbose
        if gpryacqfuncs.is_acquisition_function(acq_func):
            self.acq_func = acq_func
        elif isinstance(acq_func, (Mapping, str)):
            spec = {acq_func: {}} if isinstance(acq_func, str) else dict(acq_func)
            name = list(spec)[0]
            args = dict(spec[name] or {})
            args["dimension"] = self.n_d
            try:
                cls = getattr(gpryacqfuncs, name)
            except AttributeError as excpt:
                raise ValueError(f"Unknown AcquisitionFunction class {name}. Available: "
                                 f"{gpryacqfuncs.builtin_names()}") from excpt
            try:
                self.acq_func = cls(**args)
            except Exception as excpt:
                raise ValueError("Error when initialising the AcquisitionFunction object "
                                 f"{name} with arguments {args}: {excpt}") from excpt
        else:
            raise TypeError("acq_func should be an AcquisitionFunction or a str or dict "
                            f"specification. Got {acq_func}")

    def __call__(self, X, gpr, eval_gradient=False):
        return self.acq_func(X, gpr, eval_gradient=eval_gradient)


class BatchOptimizer(GenericGPAcquisition):
    """Batch acquisition by optimising the acquisition function from several starting points and
    appending "lies" (gpry/gp_acquisition.py:127-525): same control flow, random-number order and
    optimiser (scipy's L-BFGS-B with the analytic x-gradient) as the reference.  On the device: the optimiser
    runs of a proposal advance side by side, the posterior evaluations of a round in one ``gpry_predict_grad_batch``
    (``lockstep``; one ``gpry_predict_point`` per step with ``lockstep=False``, the reference's one-run-after-another
    form), and each lie extends the factor by a border row (``gpry_append_rows``, O(N^2)) where the reference rebuilds
    and refactorises the model (gpry/gp_acquisition.py:488-491 -> gpry/gpr.py:1015-1017, O(N^3)).

    Parity of the side-by-side form with the one-after-another loop is to the tolerance of the posterior (1e-8 of the
    mean, 1e-9 C of the variance), not to the bit: a round of four or more points goes through the batched kernels, the
    last runs through the one-point call, and the two sum in different orders -- a run's trajectory can therefore depend in
    its last digits on how many other runs are still going.  (The hyper-parameter fit's side-by-side form IS bit-identical
    to its sequential loop: ``gpry_lml_batch`` gives every theta the arithmetic of a single evaluation.)"""

    def __init__(self, bounds, preprocessing_X=None, verbose=1, acq_func="LogExp", proposer=None,
                 acq_optimizer="fmin_l_bfgs_b", n_restarts_optimizer="5d", n_repeats_propose=10, lockstep="auto"):
        super().__init__(bounds=bounds, preprocessing_X=preprocessing_X, verbose=verbose, acq_func=acq_func)
        self.obj_func = None
        # "auto" / True: the L-BFGS-B runs of one proposal are stepped side by side and their posterior evaluations batched
        # (gpry_amd.lockstep) when optimiser, acquisition function and model allow it; False: one run after another
        self.lockstep = lockstep
        if proposer is None:
            proposer = PartialProposer(self.bounds_, CentroidsProposer(self.bounds_))
        elif not isinstance(proposer, Proposer):
            raise TypeError(f"'proposer' must be a Proposer instance. Got {proposer} of type {type(proposer)}.")
        else:
            proposer.update_bounds(self.bounds_)
        self.proposer = proposer
        if acq_optimizer == "auto":
            acq_optimizer = "fmin_l_bfgs_b" if self.acq_func.hasgradient else "sampling"
        if isinstance(acq_optimizer, str):
            if acq_optimizer == "fmin_l_bfgs_b" and not self.acq_func.hasgradient:
                raise ValueError("In order to use the 'fmin_l_bfgs_b' optimizer the acquisition function needs "
                                 "to be able to return gradients. Got %s" % self.acq_func)
            if acq_optimizer not in ("fmin_l_bfgs_b", "sampling"):
                raise ValueError("Supported internal optimizers are 'auto', 'lbfgs' or 'sampling', "
                                 "got {0}".format(acq_optimizer))
        self.acq_optimizer = acq_optimizer
        self.n_restarts_optimizer = get_Xnumber(n_restarts_optimizer, "d", self.n_d, int, "n_restarts_optimizer")
        self.n_repeats_propose = n_repeats_propose
        self.mean_ = None
        self.cov = None

    def _objective(self, gpr):
        def obj_func(X, eval_gradient=False):
            X = np.expand_dims(np.asarray(X), axis=0)
            if X.ndim != 2:
                raise ValueError("X is {}-dimensional, however, it must be 2-dimensional.".format(X.ndim))
            if self.preprocessing_X is not None:
                X = self.preprocessing_X.inverse_transform(X)
            if eval_gradient:
                acq, grad = self.acq_func(X, gpr, eval_gradient=True)
                return -1 * acq, -1 * grad
            return -1 * self.acq_func(X, gpr, eval_gradient=False)
        return obj_func

    def optimize_acquisition_function(self, gpr, i, bounds=None, rng=None):
        """One optimisation run: ``i == 0`` from the last in-bounds training point, otherwise from
        the best of ``n_repeats_propose + 1`` finite proposals (gpry/gp_acquisition.py:270-389)."""
        self.proposer.update(gpr)
        use_bounds = self.bounds_ if bounds is None else bounds
        self.proposer.update_bounds(use_bounds)
        if not self.obj_func:
            if not hasattr(gpr, "X_train_"):
                raise AttributeError("The model which is given has not been fed any points.")
            self.obj_func = self._objective(gpr)
        px = self.preprocessing_X
        tbounds = px.transform_bounds(use_bounds) if px is not None else use_bounds
        x0, settled = self._starting_point(gpr, i, bounds, rng)
        if settled is not None:
            return x0, settled
        return self._constrained_optimization(self.obj_func, x0, tbounds)

    def _starting_point(self, gpr, i, bounds, rng):
        """Where run ``i`` starts, in the optimiser's coordinates: ``(x0, None)``, or ``(x0, value)`` when there is
        nothing to optimise (no finite proposal).  Consumes the generator exactly as the reference's run does."""
        px = self.preprocessing_X

        def coords(x0):
            return px.transform(x0) if px is not None else x0

        if i == 0:
            return coords(next(X for X in gpr.X_train[::-1] if np.all(is_in_bounds(X, bounds, check_shape=False)))), None
        d = self.bounds_.shape[0]
        x0s = np.empty((self.n_repeats_propose + 1, d))
        values = np.empty(self.n_repeats_propose + 1)
        ifull = 0
        x0 = value = None
        # The reference draws a proposal, evaluates it and goes on until it holds n_repeats_propose + 1 finite values (or runs
        # out of tries).  Drawn in rounds of exactly as many as are still missing, the proposals are the same ones in the same
        # order -- the last finite one of the reference's loop is the last of a round -- and a round is ONE posterior evaluation.
        tries = 10 * d * self.n_restarts_optimizer
        while tries > 0 and ifull <= self.n_repeats_propose:
            k = min(self.n_repeats_propose + 1 - ifull, tries)
            tries -= k
            batch = np.array([self.proposer.get(rng=rng) for _ in range(k)])
            vals = np.ravel(self.acq_func(batch, gpr)) if self.lockstep is not False else \
                np.array([np.ravel(self.acq_func(x, gpr))[0] for x in batch])
            x0, value = batch[-1], vals[-1:]
            for x, v in zip(batch, vals):
                if np.isfinite(v):
                    x0s[ifull], values[ifull] = x, v
                    ifull += 1
        if ifull > self.n_repeats_propose:
            return coords(x0s[np.argmax(values)]), None
        if ifull > 0:
            return coords(x0s[np.argmax(values[:ifull])]), None
        return coords(x0), -1 * value

    def _can_lockstep(self, gpr):
        if self.lockstep is False or self.acq_optimizer != "fmin_l_bfgs_b":
            return False
        if not (hasattr(self.acq_func, "value_and_grad_rows") and hasattr(gpr, "predict_with_gradients")):
            return False
        if np.iterable(gpr.noise_level) and getattr(self.acq_func, "sigma_n", None) is None:
            return False
        from gpry_amd import lockstep
        if lockstep.available():
            return True
        lockstep.warn_once("the acquisition optimiser's restarts")
        return False

    def _optimize_side_by_side(self, gpr, n_runs, use_bounds, rng, proposal_X, acq_X):
        """The ``n_runs`` optimiser runs of one proposal with their posterior evaluations batched: the starting points are
        drawn run by run first (the optimiser itself draws nothing, so the generator is consumed as in the reference), then
        the runs advance together (same routine, tolerances and stopping rules as ``fmin_l_bfgs_b``; function values equal
        to the one-after-another loop's within the posterior tolerance, see the class docstring)."""
        from gpry_amd import lockstep
        self.proposer.update(gpr)
        self.proposer.update_bounds(use_bounds)
        px = self.preprocessing_X
        tbounds = px.transform_bounds(use_bounds) if px is not None else use_bounds
        todo = []
        for i in range(n_runs):
            # (the proposer's view of the model does not change between the runs of one proposal)
            x0, settled = self._starting_point(gpr, i, use_bounds, rng)
            proposal_X[i] = x0
            if settled is None:
                todo.append(i)
            else:
                acq_X[i] = settled

        def fg(Xt):
            X = px.inverse_transform(Xt) if px is not None else Xt
            if len(X) < 4:      # the last runs still going: the one-point call is cheaper than a batched one of so few rows
                rows = [self.acq_func(x[None, :], gpr, eval_gradient=True) for x in X]
                return (-1 * np.array([np.ravel(a)[0] for a, _ in rows]), -1 * np.array([np.ravel(g) for _, g in rows]))
            acq, grad = self.acq_func.value_and_grad_rows(X, gpr)
            return -1 * acq, -1 * grad

        if todo:
            Xo, Fo, _ = lockstep.minimize_lockstep(fg, proposal_X[todo], np.asarray(tbounds, dtype=float))
            proposal_X[todo], acq_X[todo] = Xo, Fo

    def multi_add(self, gpr, n_points=1, bounds=None, rng=None, force_resample=False):
        """``n_points`` proposals, each the best of ``n_restarts_optimizer`` optimiser runs on the model
        augmented by the previous proposals at their predicted means (gpry/gp_acquisition.py:391-497)."""
        if not (isinstance(n_points, int) and n_points > 0):
            raise ValueError(f"n_points should be int > 0, got {n_points}")
        rng = get_random_generator(rng)
        use_bounds = self.bounds_ if bounds is None else bounds
        X_opts, y_lies, acq_vals = np.empty((n_points, gpr.d)), np.empty(n_points), np.empty(n_points)
        gpr_ = deepcopy(gpr)                    # the lies go into a copy
        n_runs = self.n_restarts_optimizer
        proposal_X, acq_X = np.empty((n_runs, gpr_.d)), np.empty((n_runs,))
        side_by_side = self._can_lockstep(gpr_)
        for ipoint in range(n_points):
            if side_by_side:
                self._optimize_side_by_side(gpr_, n_runs, use_bounds, rng, proposal_X, acq_X)
            else:
                for i in range(n_runs):
                    proposal_X[i], acq_X[i] = self.optimize_acquisition_function(gpr_, i, bounds=use_bounds, rng=rng)
            self.obj_func = None
            best = np.argmin(acq_X) if np.any(np.isfinite(acq_X)) else len(acq_X) - 1
            X_opt = proposal_X[best]
            if self.preprocessing_X is not None:
                X_opt = self.preprocessing_X.inverse_transform(X_opt)
            X_opt = np.array([X_opt])
            y_lie = gpr_.predict(X_opt)
            if ipoint < n_points - 1:
                lie_noise = np.array([np.mean(gpr_.noise_level)]) if np.iterable(gpr_.noise_level) else None
                gpr_.append_to_data(X_opt, y_lie, noise_level=lie_noise, fit_gpr=False, fit_classifier=False)
            X_opts[ipoint], y_lies[ipoint], acq_vals[ipoint] = X_opt[0], y_lie[0], -1 * acq_X[best]
        gpr.n_eval = gpr_.n_eval
        self.stats = {"border_updates": getattr(gpr_, "n_border_updates", 0), "side_by_side": bool(side_by_side)}
        return X_opts, y_lies, acq_vals

    def _constrained_optimization(self, obj_func, initial_X, bounds):
        if self.acq_optimizer == "fmin_l_bfgs_b":
            res = scipy.optimize.fmin_l_bfgs_b(obj_func, initial_X, args={"eval_gradient": True}, bounds=bounds,
                                               approx_grad=False)
            return res[0], res[1]
        if self.acq_optimizer == "sampling":
            res = scipy.optimize.minimize(obj_func, initial_X, args=(False), method="Powell", bounds=bounds)
            return res.x, res.fun
        if callable(self.acq_optimizer):
            return self.acq_optimizer(obj_func, initial_X, bounds=bounds)
        raise ValueError("Unknown optimizer %s." % self.acq_optimizer)


class NORA(GenericGPAcquisition):
    """Nested-sampling Optimisation for Ranked Acquisition, device sweep edition.

    Constructor arguments as gpry/gp_acquisition.py:584-601.  The external nested
    samplers (PolyChord / UltraNest / nessai) only *produce* the candidate set and are not
    part of this package: use ``sampler="uniform"``, or override ``do_MC_sample`` to
    inject a pool ``(X, None, None, weights)``.  ``comm`` (optional) is a communicator with
    ``world``, ``rank`` and ``allgather(ndarray)`` (``gpry_amd._lib.RcclComm``) to shard the
    sweep over several GPUs.  ``gather_y``: whether every rank assembles the full ``(y, sigma_y)``
    arrays of the pool after a sharded sweep (``last_MC_sample()[1:3]``; the reference merges
    them on rank 0, gpry/mpi.py:118-131) -- ``True``, ``False`` (they stay ``None``, which the
    reference's interface allows: "may be None if not computed while sampling"), or ``"auto"``
    = only when a later reweighting will need them (``mc_every > 1``).  The ranking itself never
    needs them: it works on the devices' shortlists.  ``devices``: the GPUs that share the sweep
    inside THIS process -- ``None`` (default: ``GPRY_HIP_DEVICES`` if set, else every visible GPU when
    the process is not one rank of a multi-process launch), ``"all"``, an int ``k`` (the first ``k``
    GPUs), a list of device indices (repeats = several contexts on one GPU) or a ready group object
    with the ``gpry_amd._lib.DeviceGroup`` interface.  Candidate shards, model replication and the
    shortlist merge are those of the multi-rank path; results are identical to one context's.
    """

    def __init__(self, bounds, preprocessing_X=None, verbose=1, acq_func="LogExp", sampler=None,
                 mc_every="1d", nlive_per_training=3, nlive_max="25d", nlive_per_dim_max=None,
                 num_repeats="5d", num_repeats_per_dim=None, precision_criterion_target=0.01,
                 nprior_per_nlive=10, max_ncalls=None, tmpdir=None, comm=None,
                 shortlist_size=None, gather_y="auto", devices=None):
        super().__init__(bounds=np.asarray(bounds), preprocessing_X=preprocessing_X,
                         verbose=verbose, acq_func=acq_func)
        self.log_header = f"[ACQUISITION : {self.__class__.__name__}] "
        self.mc_every = get_Xnumber(mc_every, "d", self.n_d, int, "mc_every")
        self.mc_every_i = 0
        self.tmpdir = tmpdir
        self.i = 0
        self.acq_func_y_sigma = None
        self.sampler = sampler
        self._init_nested_sampler()
        self.nlive_per_training = nlive_per_training
        self.nlive_max = (nlive_per_dim_max * self.n_d if nlive_per_dim_max is not None else
                          get_Xnumber(nlive_max, "d", self.n_d, int, "nlive_max"))
        self.num_repeats = (num_repeats_per_dim * self.n_d if num_repeats_per_dim is not None else
                            get_Xnumber(num_repeats, "d", self.n_d, int, "num_repeats"))
        self.precision_criterion_target = precision_criterion_target
        self.nprior_per_nlive = nprior_per_nlive
        self.max_ncalls = max_ncalls
        self._X_mc = self._y_mc = self._sigma_y_mc = self._w_mc = None
        self._X_mc_reweight = self._y_mc_reweight = None
        self._sigma_y_mc_reweight = self._w_mc_reweight = None
        self.is_last_MC_reweighted = None
        self.pool = None
        self.comm = comm
        if devices is not None and comm is not None and getattr(comm, "world", 1) > 1:
            raise ValueError("pass either comm= (one process per GPU) or devices= (one process, several "
                             "GPUs), not both")
        self.devices = devices
        self._group = self._group_key = self._group_model = None
        self.shortlist_size = shortlist_size
        self.gather_y = gather_y
        self._X_already_proposed = np.empty((0, self.n_d))
        self.stats = {}

    # -- bookkeeping identical in meaning to the reference --------------------------------
    @property
    def pool_size(self):
        return None if self.pool is None else len(self.pool)

    def _init_nested_sampler(self):
        s = self.sampler
        if s is None:
            s = "uniform"   # no external sampler ships with this package
        if s.lower() not in ("uniform",):
            raise NestedSamplerNotInstalledError(
                f"Nested sampler '{s}' is an external package outside this hot-path package; "
                "pass sampler='uniform' or override do_MC_sample to inject the candidate pool.")
        self.sampler = s

    def update_NS_precision(self, gpr):
        nlive = min(self.nlive_per_training * gpr.n, self.nlive_max)
        return {"nlive": nlive, "num_repeats": self.num_repeats,
                "precision_criterion": self.precision_criterion_target,
                "nprior": int(self.nprior_per_nlive * nlive), "max_ncalls": self.max_ncalls}

    def log(self, msg, level=None):
        if level is None or level <= self.verbose:
            print(self.log_header + msg)

    def do_MC_sample(self, gpr, bounds, rng=None, sampler=None):
        """Returns ``(X, y, sigma_y, weights)``; any of the last three may be None."""
        sampler = sampler or self.sampler
        if sampler.lower() == "uniform":
            return self._do_MC_sample_uniform(gpr, bounds=bounds, rng=rng)
        raise ValueError(f"Sampler '{sampler}' not known.")

    def _do_MC_sample_uniform(self, gpr, bounds=None, rng=None):
        """1000 d points drawn one at a time, as gp_acquisition.py:750-758 does."""
        b = self.bounds_ if bounds is None else bounds
        rng = get_random_generator(rng)
        n_total = 1000 * gpr.d
        X = np.empty((n_total, gpr.d))
        for i in range(n_total):
            X[i] = rng.uniform(b[:, 0], b[:, 1])
        return X, None, None, None

    # -- the sweep -----------------------------------------------------------------------------
    def _shard(self, M):
        if self.comm is None or self.comm.world == 1:
            return 0, M
        w, r = self.comm.world, self.comm.rank
        per = -(-M // w)
        return min(r * per, M), min((r + 1) * per, M)

    def _resolve_devices(self):
        """Device indices of the in-process group, or None for the plain one-context sweep."""
        from gpry_amd import _lib
        spec = self.devices
        if spec is None:
            spec = os.environ.get("GPRY_HIP_DEVICES", "")
            if spec == "":
                if self.comm is not None or int(os.environ.get("WORLD_SIZE", "1")) > 1:
                    return None                     # one process per GPU: this process has its own
                spec = "all"
        if isinstance(spec, str):
            spec = spec.strip().lower()
            if spec in ("", "none", "1"):
                return None
            spec = "all" if spec == "all" else [int(v) for v in spec.split(",")]
        if isinstance(spec, str):                   # "all"
            n = _lib.device_count()
            return list(range(n)) if n > 1 else None
        if isinstance(spec, (int, np.integer)):
            return list(range(int(spec))) if spec > 1 else None
        spec = [int(v) for v in spec]
        return spec if len(spec) > 1 else None

    def _sweeper(self, gpr):
        """What runs the sweep: the model's own context, or the device group that shards it inside
        this process (built on first use around the model's context as member 0)."""
        if hasattr(self.devices, "sweep_topk"):         # a ready group (tests inject a double)
            return self.devices, True
        if not hasattr(gpr.device, "_h"):               # not a libgpry_hip context
            return gpr.device, False
        if self._group is not None and self._group_key is gpr.device:
            return self._group, True
        devs = self._resolve_devices()
        if devs is None:
            return gpr.device, False
        from gpry_amd import _lib
        if devs[0] != gpr.device.device:                # member 0 is the model's own context
            devs = [gpr.device.device] + [v for v in devs if v != gpr.device.device]
        self._group = _lib.DeviceGroup(devs, adopt=gpr.device)
        self._group_key, self._group_model = gpr.device, None
        return self._group, True

    def _device_sweep(self, gpr, X, need_arrays=False):
        """Mean, std and LogExp acquisition of every row of X (this rank's shard on the
        device; y and sigma all-gathered so that every rank holds the full arrays)."""
        t0 = time()
        X = np.ascontiguousarray(X, dtype=float)
        M = len(X)
        gpr._ensure_factor()
        gpr._push_affine()
        dev, grouped = self._sweeper(gpr)
        lo, hi = (0, M) if grouped else self._shard(M)
        if grouped:
            # replicate the model on the other members when it has changed since the last sweep
            key = (gpr.device, getattr(gpr, "_factor_epoch", None))
            if key[1] is None or key != self._group_model:
                kid, theta = gpr._device_theta()
                info = dev.set_model(gpr.X_train_, gpr.y_train_, gpr.alpha, kid, theta, gpr._affine_args())
                if info:
                    raise np.linalg.LinAlgError(f"a group member could not factorise the model (info={info})")
                self._group_model = key
        # classifier / trust-region verdicts: on the device if they have a device form
        mask = None
        if hi > lo:
            on_device = hasattr(gpr, "_push_gates") and (gpr._push_gates(sinks=[gpr.device, dev]) if grouped
                                                         else gpr._push_gates())
            if not on_device:
                mask = gpr._masks(X[lo:hi], False, False)
        noise = gpr.noise_level
        if np.iterable(noise):
            raise ValueError("NORA needs a scalar noise_level (the reference passes it raw to "
                             "LogExp.f, gp_acquisition.py:1049-1051)")
        # the very same array object as last time is still resident in HBM: skip the upload
        resident = (X is getattr(self, "_sweep_X", None) and getattr(self, "_sweep_dev", None) is dev
                    and (lo, hi) == (self._sweep_lo, self._sweep_hi))
        sharded = self.comm is not None and self.comm.world > 1
        gather = (self.mc_every > 1) if self.gather_y == "auto" else bool(self.gather_y)
        # one rank: y / sigma stay on the device and are fetched when somebody asks for them
        # (last_MC_sample, a later reweighting) -- 16 MB of copies per 1e6 candidates otherwise
        lazy = (not sharded and not need_arrays and self.gather_y == "auto"
                and hasattr(dev, "sweep_fetch"))
        want = ("y", "sigma") if ((gather and sharded) or (not sharded and not lazy)) else ()
        out = dev.sweep_logexp(None if resident else X[lo:hi], self.acq_func.zeta, gpr.y_max,
                               noise, mask=mask, M=hi - lo, want=want)
        self._sweep_dev = dev
        gpr.n_eval += M
        if out["n_nan"]:
            raise ValueError("Acquisition function value not a number: nan")
        y, s = (out.get("y"), out.get("sigma")) if want else (None, None)
        if sharded and gather:
            per = -(-M // self.comm.world)
            buf = np.zeros((2, per))
            buf[0, :hi - lo], buf[1, :hi - lo] = y, s
            allb = self.comm.allgather(buf)
            if per * self.comm.world == M:       # even shards: the gathered rows are the arrays
                y, s = allb[:, 0, :].reshape(-1), allb[:, 1, :].reshape(-1)
            else:
                y = np.concatenate([allb[r, 0, :max(0, min(per, M - r * per))] for r in range(self.comm.world)])
                s = np.concatenate([allb[r, 1, :max(0, min(per, M - r * per))] for r in range(self.comm.world)])
        self._lazy = (dev, dev.sweep_epoch) if lazy else None
        self._sweep_X, self._sweep_lo, self._sweep_hi = X, lo, hi
        self._sweep_grouped = grouped
        self.stats["sweep_s"] = time() - t0
        self.stats["sweep_M"] = M
        self.stats["sweep_contexts"] = getattr(dev, "size", 1) if grouped else 1
        # which form of the cross-kernel panel the model's error estimates allowed (gpry_sweep_info): a fitted model that
        # drops from the matrix-pipe form to the difference form sweeps ~1 % slower -- say so once instead of silently
        info_dev = dev.member(0) if grouped and hasattr(dev, "member") else dev
        if hasattr(info_dev, "sweep_info"):
            info = info_dev.sweep_info()
            self.stats.update(info)
            # (estimates of 0: the gate was never evaluated -- Matern-1/2 always takes the difference form, option "cross_mfma" = 0)
            if info["panel_form"] == "difference" and info["panel_error_estimate"] > 0 and \
                    not getattr(NORA, "_warned_panel_form", False):
                NORA._warned_panel_form = True
                warnings.warn("gpry_amd: the cross-kernel panel of the sweep is built in the difference form for this model "
                              f"(error estimates of the matrix-pipe form: mean {info['panel_error_estimate']:.2g}, variance "
                              f"{info['panel_error_variance']:.2g} against a gate of {info['panel_gate']:.2g}); "
                              "see acq.stats['panel_form']", RuntimeWarning)
        return y, s

    def _shortlist(self, gpr, K, exclude_global):
        """Global descending stream prefix: records (acq, y, sigma, idx) valid down to
        ``bound`` (every candidate not returned has acq <= bound)."""
        lo, hi = self._sweep_lo, self._sweep_hi
        excl = None
        if exclude_global is not None and len(exclude_global):
            e = np.asarray(exclude_global, dtype=np.int64)
            excl = e[(e >= lo) & (e < hi)] - lo
        # K is the length wanted for the MERGED list: shards of one pool are statistically alike, so
        # nearly every record of every member survives the hold-back rule and 2K / members each is
        # plenty (the caller extends the list if it was not: exactness never depends on this choice)
        if getattr(self, "_sweep_grouped", False):
            # the group merges its members' shortlists itself (same rule as below, in the library)
            return self._sweep_dev.sweep_topk(-(-2 * K // self._sweep_dev.size), exclude=excl)
        if self.comm is not None and self.comm.world > 1:
            K = -(-2 * K // self.comm.world)
        top, bound = gpr.device.sweep_topk(K, exclude=excl)
        top = top.copy()
        top["idx"] += lo
        if self.comm is None or self.comm.world == 1:
            return top, bound, len(top) < K
        from gpry_amd._lib import CAND_DTYPE
        rec = np.zeros(K + 1, dtype=CAND_DTYPE)
        rec[:len(top)] = top
        rec[len(top):]["idx"] = -1
        rec[K]["acq"], rec[K]["idx"] = bound, len(top)   # trailer: this rank's bound and count
        allr = self.comm.allgather(rec)
        parts, gbound, exhausted = [], -np.inf, True
        for r in range(self.comm.world):
            n_r = int(allr[r][K]["idx"])
            parts.append(allr[r][:n_r])
            gbound = max(gbound, float(allr[r][K]["acq"]))
            exhausted = exhausted and n_r < K
        merged = np.concatenate(parts)
        order = np.lexsort((-merged["idx"], -merged["acq"]))
        merged = merged[order]
        # entries at or below the largest per-rank bound may be preceded by candidates that a
        # rank did not send: hold them back
        valid = merged["acq"] > gbound
        if exhausted:
            valid[:] = True
        return merged[valid], (gbound if not exhausted else -np.inf), exhausted

    def _fetch_lazy(self):
        """Materialise ``_y_mc`` / ``_sigma_y_mc`` of the last (un-reweighted) sweep if they were
        left on the device and are still there."""
        lazy = getattr(self, "_lazy", None)
        if lazy is None or self._y_mc is not None:
            return
        dev, epoch = lazy
        if dev.sweep_epoch != epoch:
            raise RuntimeError("the sweep arrays of the last MC sample are no longer on the device")
        out = dev.sweep_fetch(("y", "sigma"))
        self._y_mc, self._sigma_y_mc = out["y"], out["sigma"]
        self._lazy = None

    def _set_MC_sample(self, X, y, sigma_y, w, ensure_y_sigma_y=False, gpr=None):
        """gp_acquisition.py:858-873; the (y, sigma) evaluation is the device sweep."""
        self.is_last_MC_reweighted = False
        self._X_mc, self._y_mc, self._sigma_y_mc, self._w_mc = X, y, sigma_y, w
        if ensure_y_sigma_y:
            if y is not None or sigma_y is not None:
                raise NotImplementedError("samplers that return their own y / sigma_y are not "
                                          "supported by the device sweep (all of the reference's "
                                          "samplers return None for both)")
            self._y_mc, self._sigma_y_mc = self._device_sweep(gpr, X)
            self._dev_index = np.arange(len(X))

    def _reweight_last_MC_sample(self, gpr, bounds=None, ensure_sigma_y=False):
        """gp_acquisition.py:875-919."""
        if not self.is_last_MC_reweighted:
            self._fetch_lazy()          # the old y is needed below; the new sweep overwrites it
        self.is_last_MC_reweighted = True
        if self._X_mc is None:
            raise ValueError("No samples yet!")
        if self._y_mc is None:
            raise ValueError("Original logp was not stored. Cannot reweight!")
        Xr = np.copy(self._X_mc)
        within = None
        if bounds is not None:
            within = is_in_bounds(Xr, bounds, check_shape=False)
            Xr = Xr[within]
        yr, sr = self._device_sweep(gpr, Xr, need_arrays=True)
        with np.errstate(all="ignore"):
            y_old, w_old = self._y_mc, self._w_mc
            if within is not None:
                y_old = y_old[within]
                w_old = w_old[within] if w_old is not None else None
            factor = np.exp(yr - y_old)
            w_new = (w_old if w_old is not None else np.ones(Xr.shape[0])) * factor
            w_new /= max(w_new)
        keep = np.where(w_new != 0)[0]
        self._dev_index = keep          # positions (device order) of the samples that survive
        self._dropped = np.where(w_new == 0)[0]
        self._w_mc_reweight, self._X_mc_reweight, self._y_mc_reweight, \
            self._sigma_y_mc_reweight = remove_0_weight_samples(w_new, Xr, yr, sr)

    def last_MC_sample(self, copy=False, warn_reweight=True):
        if self.is_last_MC_reweighted:
            if warn_reweight:
                warnings.warn("This is a reweighted sample! (disable with `warn_reweight=False`)")
            vals = (self._X_mc_reweight, self._y_mc_reweight, self._sigma_y_mc_reweight,
                    self._w_mc_reweight)
        else:
            self._fetch_lazy()
            vals = (self._X_mc, self._y_mc, self._sigma_y_mc, self._w_mc)
        if copy:
            vals = tuple(None if v is None else np.copy(v) for v in vals)
        return vals

    @property
    def mean(self):
        Xs, _, _, ws = self.last_MC_sample(copy=False, warn_reweight=False)
        return np.average(Xs.T, weights=ws, axis=-1)

    @property
    def cov(self):
        Xs, _, _, ws = self.last_MC_sample(copy=False, warn_reweight=False)
        return np.cov(Xs.T, aweights=ws, ddof=0)

    # -- the acquisition step ------------------------------------------------------------------
    def multi_add(self, gpr, n_points=1, bounds=None, rng=None, force_resample=False):
        """Propose ``n_points`` locations by Kriging-believer ranking of the candidate pool
        (gp_acquisition.py:971-1108).  Returns ``(X, y_lies, acq)`` with unconditioned acq."""
        if not (isinstance(n_points, (int, np.integer)) and n_points > 0):
            raise ValueError(f"n_points should be int > 0, got {n_points}")
        n_points = int(n_points)
        rng = get_random_generator(rng)
        t_start = time()
        resample = not bool(self.mc_every_i % self.mc_every) or force_resample
        if resample:
            self._set_MC_sample(*self.do_MC_sample(gpr, bounds=bounds, rng=rng),
                                ensure_y_sigma_y=True, gpr=gpr)
            self._X_already_proposed = np.empty((0, gpr.d))
            self._dropped = np.empty(0, dtype=np.int64)
        else:
            self._reweight_last_MC_sample(gpr, bounds=bounds, ensure_sigma_y=True)
        self.mc_every_i += 1
        X_dev = self._sweep_X       # rows in device order (superset of last_MC_sample's rows)
        # rows to ignore: weight-0 samples of a reweighted set and points proposed since the
        # last resampling (:1037-1047; both samples assumed unique, as there)
        exclude = list(self._dropped)
        if len(self._X_already_proposed):
            # first row of the pool equal to each proposed point; one pass over the first
            # coordinate narrows 1e6 rows down to the handful worth comparing in full
            near = np.flatnonzero(np.isin(X_dev[:, 0], self._X_already_proposed[:, 0]))
            for row in self._X_already_proposed:
                hit = near[np.all(X_dev[near] == row, axis=1)] if near.size else near
                if hit.size:
                    exclude.append(int(hit[0]))
        exclude = np.unique(np.asarray(exclude, dtype=np.int64))
        self.acq_func_y_sigma = partial(self.acq_func.f, baseline=gpr.y_max,
                                        noise_level=gpr.noise_level, zeta=self.acq_func.zeta)
        self.log(f"({(time() - t_start):.2g} sec) " +
                 (f"Obtained new MC sample with {self.sampler}" if resample
                  else "Re-evaluated previous MC sample"), level=2)
        t_rank = time()
        self.pool = RankedPool(n_points, gpr=gpr, acq_func=self.acq_func_y_sigma,
                               verbose=self.verbose - 3)
        K = int(self.shortlist_size or max(256, 16 * n_points))
        fed = 0
        with np.errstate(divide="ignore"):
            while True:
                top, bound, exhausted = self._shortlist(gpr, K, exclude)
                new = top[fed:]
                if len(new):
                    self.pool.add(X_dev[new["idx"]], new["y"].copy(), new["sigma"].copy(),
                                  new["acq"].copy(), method="single sort acq")
                fed = len(top)
                # every candidate not fed has acq <= bound: if that is at or below the pool's
                # admission level they would all return at the early-out (:1432)
                if exhausted or self.pool.min_acq >= bound:
                    break
                K *= 4
        self.stats.update(shortlist=fed, cache_models=self.pool.cache_counter,
                          rank_s=time() - t_rank)
        merged = self.pool.copy(drop_empty=True)
        X_pool, y_pool = merged.X[:n_points], merged.y[:n_points]
        with np.errstate(divide="ignore"):
            acq_pool = self.acq_func_y_sigma(y_pool, merged.sigma[:n_points])
        self._X_already_proposed = np.concatenate([self._X_already_proposed, X_pool])
        self.pool.reset_cache()
        self.log(f"({(time() - t_rank):.2g} sec) Ranked pool of candidates.", level=2)
        return X_pool, y_pool, acq_pool


class RankedPool:
    """Ranked pool of proposals under the Kriging-believer rule (gp_acquisition.py:1194).

    Slot ``i`` holds the candidate whose acquisition value, evaluated with the standard
    deviation of the GP augmented by slots ``0..i-1`` at their predicted means, is the
    largest among the remaining ones.  ``acq_cond == -inf`` marks an empty slot; the
    arrays carry one spare slot at the end.
    """

    def __init__(self, size, gpr, acq_func, verbose=1):
        self._gpr = gpr
        self._acq_func = acq_func
        self.verbose = verbose
        self.X = np.zeros((size + 1, gpr.d))
        self.y = np.zeros(size + 1)
        self.acq_cond = np.full(size + 1, -np.inf)
        self.sigma = np.zeros(size + 1)
        self.acq = np.zeros(size + 1)
        self.reset_cache()
        self.cache_counter = 0

    def __len__(self):
        return len(self.y) - 1

    @property
    def min_acq(self):
        """Admission level: conditioned acquisition of the last real slot (-inf if not full)."""
        return self.acq_cond[len(self) - 1]

    def log(self, level=None, msg=""):
        if level is None or level <= self.verbose:
            print(msg)

    def str_point(self, X, y, sigma, acq, sigma_cond=None, acq_cond=None):
        s_c = f" (cond: {sigma_cond})" if sigma_cond is not None else ""
        a_c = f" (cond: {acq_cond})" if acq_cond is not None else ""
        return f"{X}, y = {y} +/- {sigma}{s_c}; acq = {acq}{a_c}"

    def __str__(self):
        return "\n".join(f"{i + 1} : " + self.str_point(self.X[i], self.y[i], self.sigma[i],
                                                        self.acq[i], acq_cond=self.acq_cond[i])
                         for i in range(len(self)))

    # -- conditioned models ------------------------------------------------------------------------
    def cache_model(self, i):
        """Model = base GP + slots 0..i (gp_acquisition.py:1522-1555), as a bordered factor."""
        if i < 0:
            return self._gpr
        self.gpr_cond[i] = self._gpr.conditioned(self.X[:i + 1], self.y[:i + 1])
        self.cache_counter += 1
        return self.gpr_cond[i]

    def reset_cache(self):
        self.gpr_cond = [None] * len(self.X)

    # -- filling -------------------------------------------------------------------------------------
    def add(self, X, y=None, sigma=None, acq=None, method="single sort acq"):
        """Offer points to the pool (gp_acquisition.py:1290-1335)."""
        X = np.atleast_2d(X)
        y = None if y is None else np.atleast_1d(y)
        sigma = None if sigma is None else np.atleast_1d(sigma)
        if y is None:
            y, sigma = self._gpr.predict(X, return_std=True, validate=False)
        elif sigma is None:
            sigma = self._gpr.predict_std(X, validate=False)
        if acq is None:
            acq = self._acq_func(y, sigma)
        m = method.lower()
        if hasattr(self._gpr, "kb_session") and 0 < len(X) <= 16384:
            # one device pass computes u(x) = V k*(x) for the whole offer; conditioned
            # standard deviations are then look-ups (gpry_amd/kriging.py)
            self._gpr.kb_session().register(X)
        if m == "bulk":
            return self.add_bulk(X, y, sigma, acq)
        if not m.startswith("single"):
            raise ValueError(f"Algorithm '{method}' not known.")
        order = range(len(X))
        by_acq = False
        if "sort" in m:
            key = m.split()[-1]
            order = np.argsort({"acq": acq, "y": y}[key])[::-1]
            by_acq = key == "acq"
        for i in order:
            if by_acq and acq[i] <= self.min_acq:
                # descending acquisition: this and all later offers return at add_one's
                # first test (:1432) without touching the pool
                break
            self.add_one(X[i], y[i], sigma[i], acq[i])

    def add_bulk(self, X, y, sigma, acq, i_start=0):
        """Fill slot after slot from the whole batch (gp_acquisition.py:1337-1390)."""
        X, y, sigma, acq = np.atleast_2d(X), np.asarray(y), np.asarray(sigma), np.asarray(acq)
        slot = i_start
        while True:
            if slot == 0:
                a_cond = acq
            else:
                model = self.cache_model(slot - 1)
                a_cond = self._acq_func(y, model.predict_std(X, validate=False))
            if a_cond.size == 0:
                return
            best = int(np.argmax(a_cond))
            if a_cond[best] == np.inf:
                return
            self.X[slot], self.y[slot] = X[best], y[best]
            self.sigma[slot], self.acq[slot] = sigma[best], acq[best]
            self.acq_cond[slot] = a_cond[best]
            if slot == len(self) - 1:
                return
            keep = a_cond != -np.inf
            keep[best] = False
            X, y, sigma, acq = X[keep], y[keep], sigma[keep], acq[keep]
            slot += 1

    def _position_for(self, a_cond):
        """First slot, scanning upwards from the bottom, whose conditioned value is >= a_cond
        decides the provisional rank (gp_acquisition.py:1470-1474)."""
        n = len(self)
        for up in range(n):
            if self.acq_cond[n - 1 - up] >= a_cond:
                return n - up
        return 0

    def add_one(self, X, y=None, sigma=None, acq=None, acq_nan_is_null=False):
        """Offer one point (gp_acquisition.py:1392-1520)."""
        if acq is not None and acq <= self.min_acq:
            return
        X = np.atleast_2d(X)
        if y is None:
            y, sigma = self._gpr.predict(X, return_std=True, validate=False)
            y, sigma = y[0], sigma[0]
        if sigma is None:
            sigma = self._gpr.predict_std(X, validate=False)
        if acq is None:
            acq = self._acq_func(y, sigma)
        if acq <= self.min_acq:
            return
        if np.isnan(acq):
            if not acq_nan_is_null:
                raise ValueError(f"Acquisition function value not a number: {acq}")
            acq = -np.inf
        n = len(self)
        a_cond = deepcopy(acq)
        last_pos = n
        while True:
            pos = self._position_for(a_cond)
            if pos in (0, last_pos, n):
                break
            s_cond = self.gpr_cond[pos - 1].predict_std(X, validate=False)[0]
            a_cond = min(a_cond, self._acq_func(y, s_cond))   # conditioning cannot raise it
            last_pos = pos
        if pos >= n:
            return
        for arr, val in ((self.X, X), (self.y, y), (self.sigma, sigma), (self.acq, acq),
                         (self.acq_cond, a_cond)):
            arr[pos + 1:] = arr[pos:-1]
            arr[pos] = val
        assert self.acq_cond[pos] > -np.inf
        self.sort(pos + 1)
        self.acq_cond[-1] = -np.inf

    def sort(self, i_start=0):
        """Re-rank slots ``i_start..`` given that the ones above are final
        (gp_acquisition.py:1598-1670)."""
        slot = i_start
        while slot < len(self):
            upper = self.cache_model(slot - 1)
            if self.acq_cond[slot] == -np.inf:
                return
            empties = np.flatnonzero(self.acq_cond == -np.inf)
            end = int(empties[0]) if len(empties) else len(self) + 1
            s_cond = upper.predict_std(self.X[slot:end], validate=False)
            ceiling = np.inf if slot == 0 else self.acq_cond[slot - 1]
            a_cond = np.minimum(self._acq_func(self.y[slot:end], s_cond), ceiling)
            rank = np.argsort(-a_cond)
            if a_cond[rank[0]] == -np.inf:
                self.acq_cond[slot:end] = -np.inf
                return
            src = slot + rank
            self.X[slot:end] = self.X[src]
            self.y[slot:end] = self.y[src]
            self.sigma[slot:end] = self.sigma[src]
            self.acq[slot:end] = self.acq[src]
            self.acq_cond[slot:end] = a_cond[rank]
            slot += 1

    # -- copies ------------------------------------------------------------------------------------------
    def __getstate__(self):
        return deepcopy(self).__dict__

    def __deepcopy__(self, memo=None):
        new = self.__class__.__new__(self.__class__)
        new.__dict__ = {k: deepcopy(v) for k, v in self.__dict__.items()
                        if k not in ("_gpr", "_acq_func", "gpr_cond")}
        return new

    def copy(self, drop_empty=False):
        c = deepcopy(self)
        if drop_empty:
            empties = np.flatnonzero(c.acq_cond[:-1] == -np.inf)
            if len(empties):
                k = int(empties[0])
                c.X, c.y, c.acq_cond, c.sigma, c.acq = (c.X[:k], c.y[:k], c.acq_cond[:k],
                                                        c.sigma[:k], c.acq[:k])
        return c
