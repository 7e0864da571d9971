"""Multi-GPU counterparts of the reference's MPI helpers for the hot path (one process per GPU).

* ``split_number_for_parallel_processes`` -- gpry/mpi.py:80-102.
* ``get_random_generator`` -- per-rank streams from ``SeedSequence.spawn`` (gpry/mpi.py:32-50).
* ``fit_gpr_parallel`` -- the restart farm of ``Runner._fit_gpr_parallel``
  (gpry/run.py:1238-1293): the L-BFGS-B restarts are independent units, so they are split over the
  ranks (lower ranks first), every rank optimises its share against its own device, and ONE
  all-gather of ``(lml, theta)`` (2+d doubles per rank) replaces the reference's
  ``allgather((rank, lml))`` + ``bcast(pickled gpr)`` (gpry/run.py:1286-1288, ``_share_gpr``):
  every rank already holds the training set, so only the winning theta travels and each rank
  refactorises on its own GPU.

``comm`` is a ``gpry_amd._lib.RcclComm`` (or anything with ``world``, ``rank`` and
``allgather(ndarray) -> ndarray[world, ...]``); ``None`` means a single process.
"""
import numpy as np

__all__ = ["split_number_for_parallel_processes", "get_random_generator", "fit_gpr_parallel"]


def split_number_for_parallel_processes(n, n_proc=1):
    """Tasks per process, lower ranks preferred: 5 tasks on 3 processes -> [2, 2, 1]
    (gpry/mpi.py:80-102)."""
    n, n_proc = int(n), int(n_proc)
    base, extra = divmod(n, n_proc)
    return np.array([base + (1 if r < extra else 0) for r in range(n_proc)], dtype=int)


def get_random_generator(seed=None, comm=None):
    """Generator of this rank: child ``rank`` of ``SeedSequence(seed).spawn(world)``
    (gpry/mpi.py:32-50, where rank 0 spawns and scatters).  With ``seed=None`` rank 0's fresh
    entropy is shared first so that the streams are still children of one sequence."""
    if isinstance(seed, np.random.Generator):
        return seed
    world = 1 if comm is None else comm.world
    rank = 0 if comm is None else comm.rank
    if seed is None and world > 1:
        ent = np.array(np.random.SeedSequence().generate_state(4, np.uint64), dtype=np.uint64)
        seed = [int(v) for v in comm.allgather(ent)[0]]
    return np.random.default_rng(np.random.SeedSequence(seed).spawn(world)[rank])


def fit_gpr_parallel(gpr, new_X, new_y, comm=None, fit="full", n_restarts=None,
                     hyperparameter_bounds=None, fit_classifier=True):
    """Append ``(new_X, new_y)`` on every rank and fit the hyper-parameters with the restarts
    farmed out over the ranks (gpry/run.py:1238-1293).

    ``fit``: ``"full"`` -- ``n_restarts`` (default ``gpr.n_restarts_optimizer``) split over the
    ranks; ``"simple"`` -- one run per rank; ``None``/``False`` -- theta kept.  As in the
    reference only rank 0 starts from the current theta (``start_from_current =
    mpi.is_main_process``, run.py:1251); the other ranks draw their starts from their own
    ``gpr.random_state`` stream (an int / None ``random_state`` is replaced by this rank's child of
    ``SeedSequence(seed).spawn(world)`` first, so that the ranks do not optimise duplicates).  Ties are broken towards the lowest rank (``np.argmax``,
    run.py:1287).  Returns ``(best_lml, best_rank, lml_per_rank)``; afterwards every rank's
    ``gpr`` holds the winning theta and a factor for it.
    """
    world = 1 if comm is None else comm.world
    rank = 0 if comm is None else comm.rank
    if world > 1 and (gpr.random_state is None or isinstance(gpr.random_state, (int, np.integer))):
        # an int seed (or None) would give every rank the same start points: each rank gets its own
        # child stream of the seed, as Runner does (gpry/run.py:321,756 -> mpi.get_random_generator)
        gpr.set_random_state(get_random_generator(gpr.random_state, comm))
    if world > 1 and getattr(gpr, "fit_devices", None) is None:
        # one process per GPU: this rank's share of the restarts stays on this rank's GPU (a single process
        # spreads its contexts over every visible GPU, gpry_amd.gpr.fit_context_devices)
        from gpry_amd.gpr import fit_contexts
        gpr.fit_devices = [getattr(gpr.device, "device", 0)] * fit_contexts()
    if fit == "full":
        total = gpr.n_restarts_optimizer if n_restarts is None else n_restarts
        n_mine = int(split_number_for_parallel_processes(total, world)[rank])
    elif fit == "simple":
        n_mine = 1
    elif fit in (None, False, "none"):
        n_mine = 0
    else:
        raise ValueError(f"fit must be 'full', 'simple' or None, got {fit!r}")
    if n_mine:
        kwargs = {"hyperparameter_bounds": hyperparameter_bounds,
                  "start_from_current": rank == 0, "n_restarts": n_mine}
        gpr.append_to_data(new_X, new_y, fit_classifier=fit_classifier, fit_gpr=kwargs)
        lml = float(gpr.log_marginal_likelihood_value_)
    else:
        # ranks without a run still need the new rows (the reference ships them the whole
        # pickled model afterwards); theta untouched
        gpr.append_to_data(new_X, new_y, fit_classifier=fit_classifier, fit_gpr=False)
        held = getattr(gpr, "log_marginal_likelihood_value_", None)
        lml = -np.inf if (fit in ("full", "simple") or held is None) else float(held)
    theta = np.asarray(gpr.kernel_.theta, dtype=float)
    if comm is None:
        return lml, 0, np.array([lml])
    rec = comm.allgather(np.concatenate(([lml], theta)))
    lmls = np.array(rec[:, 0])
    if np.isnan(lmls).all():
        raise FloatingPointError("every rank returned a NaN log-marginal-likelihood")
    best = int(np.nanargmax(lmls)) if np.isnan(lmls).any() else int(np.argmax(lmls))
    if n_mine or fit in ("full", "simple"):
        if best != rank and not np.array_equal(rec[best, 1:], theta):
            gpr.kernel_.theta = rec[best, 1:]
            gpr._invalidate()
            gpr._ensure_factor()
        gpr.log_marginal_likelihood_value_ = float(lmls[best])
        gpr._fitted = True
    return float(lmls[best]), best, lmls
