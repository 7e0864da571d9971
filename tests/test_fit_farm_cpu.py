"""CPU tests of the host side of the fit path: the GaussianProcessRegressor mirror driven through
an oracle-backed device double (tests/oracle_device.py) against the reference's golden vectors,
and the restart farm (gpry_amd/parallel.py, gpry/run.py:1238-1293) on two gloo ranks."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import load_golden
from oracle_device import attach

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SPEC = {0: "RBF", 1: {"Matern": {"nu": 0.5}}, 2: {"Matern": {"nu": 1.5}},
               3: {"Matern": {"nu": 2.5}}}


def make_gpr(bounds, kid, **kw):
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
    kw.setdefault("account_for_inf", None)
    gpr = GaussianProcessRegressor(kernel=KERNEL_SPEC[kid], bounds=bounds,
                                   preprocessing_X=Normalize_bounds(bounds),
                                   preprocessing_y=Normalize_y(), **kw)
    return attach(gpr)


@pytest.mark.parametrize("kid,N", [(0, 48), (3, 60)])
def test_host_fit_logic_reproduces_reference_fits(kid, N):
    """F6: restart schedule, RNG order and optimiser calls of fit_gpr_hyperparameters -- with
    the oracle's arithmetic under the host code the reference's optimum is hit closely."""
    g = load_golden("fit")
    p = f"f6_k{kid}_"
    gpr = make_gpr(g[p + "bounds"], kid, n_restarts_optimizer=4, random_state=3)
    X, y, Xc = g[p + "X"], g[p + "y"], g[p + "Xc"]
    gpr.append_to_data(X[:N], y[:N], fit_gpr=True)
    assert gpr.fitted and gpr.n == N
    assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_full"]) < 1e-5
    np.testing.assert_allclose(gpr.kernel_.theta, g[p + "theta_full"], rtol=1e-3, atol=1e-3)
    m, s = gpr.predict(Xc, return_std=True)
    np.testing.assert_allclose(m, g[p + "mean_full"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s, g[p + "std_full"], rtol=1e-4, atol=1e-5)
    gpr.append_to_data(X[N:], y[N:], fit_gpr="simple")
    assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_simple"]) < 1e-4
    m, s = gpr.predict(Xc, return_std=True)
    np.testing.assert_allclose(m, g[p + "mean_simple"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(s, g[p + "std_simple"], rtol=1e-3, atol=1e-4)


def test_concurrent_restarts_select_the_sequential_optimum(monkeypatch):
    """The optimiser runs of a fit are shared by several device contexts / host threads
    (GaussianProcessRegressor._concurrent_restarts); the start points, every run and the selected
    optimum are those of the reference's sequential loop (GPRY_HIP_FIT_CONTEXTS=1)."""
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "0")      # the thread farm is under test, not the side-by-side runs
    g = load_golden("fit")
    p = "f6_k3_"
    X, y = g[p + "X"], g[p + "y"]
    out = {}
    for n_ctx in ("1", "3", "2"):
        monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", n_ctx)
        gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=5, random_state=7)
        gpr.append_to_data(X[:60], y[:60], fit_gpr=True)
        assert len(gpr._fit_devs) == int(n_ctx) - 1
        out[n_ctx] = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike,
                      gpr.predict(g[p + "Xc"]))
    for n_ctx in ("3", "2"):
        np.testing.assert_array_equal(out[n_ctx][0], out["1"][0])
        assert out[n_ctx][1] == out["1"][1] and out[n_ctx][2] == out["1"][2]
        np.testing.assert_array_equal(out[n_ctx][3], out["1"][3])


def test_restarts_stepped_side_by_side_select_the_sequential_optimum(monkeypatch):
    """Small training sets: the optimiser runs of a fit advance together and every round is ONE batched objective call
    (``_restarts_side_by_side`` / ``gpry_lml_batch``): start points, every run, the evaluation count and the selected
    optimum are those of the sequential loop."""
    g = load_golden("fit")
    p = "f6_k3_"
    X, y = g[p + "X"], g[p + "y"]
    out = {}
    for mode in ("sequential", "side by side"):
        monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
        monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "0" if mode == "sequential" else "1")
        gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=5, random_state=7)
        gpr.append_to_data(X[:60], y[:60], fit_gpr=True)
        assert bool(getattr(gpr, "fit_stats", {}).get("side_by_side")) == (mode == "side by side")
        out[mode] = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike,
                     gpr.predict(g[p + "Xc"]), gpr._rng.random())
    a, b = out["sequential"], out["side by side"]
    np.testing.assert_array_equal(b[0], a[0])
    assert b[1] == a[1] and b[2] == a[2] and b[4] == a[4]
    np.testing.assert_array_equal(b[3], a[3])


def test_side_by_side_runs_dealt_out_over_independent_groups_give_the_same_fit(monkeypatch):
    """Above 128 points the runs of a side-by-side fit are dealt out (run i to group i mod k) over k independent groups, each
    with its own lock-step driver, host thread and device context (``batch_contexts``): every run is still evaluated with
    the arithmetic of single evaluations, so the fit is the one-group fit bit for bit, whatever k; a group never holds fewer
    than three runs."""
    g = load_golden("fit_mid")
    p = "f6b_k0_"
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "1")
    out = {}
    for k in ("1", "3", "2"):
        monkeypatch.setenv("GPRY_HIP_FIT_BATCH_CONTEXTS", k)
        monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", k)
        gpr = make_gpr(g[p + "bounds"], 0, n_restarts_optimizer=7, random_state=3)
        gpr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=True)
        want = min(int(k), 7 // 3)
        assert gpr.fit_stats["side_by_side"] and gpr.fit_stats["contexts"] == want
        assert sum(gpr.fit_stats["evals_per_context"]) == sum(gpr.fit_stats["evals_per_run"])
        out[k] = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike,
                  [dv.n_lml for _, dv in gpr._fit_devs])
    for k in ("3", "2"):
        np.testing.assert_array_equal(out[k][0], out["1"][0])
        assert out[k][1] == out["1"][1] and out[k][2] == out["1"][2]
        assert len(out[k][3]) == 1 and out[k][3][0] > 0          # the second context did evaluate
    # the default width follows the size of the training set
    from gpry_amd import gpr as G
    monkeypatch.delenv("GPRY_HIP_FIT_BATCH_CONTEXTS")
    assert G.batch_contexts(100) == 1 and G.batch_contexts(300) == 1 and G.batch_contexts(1000) == 3


def test_side_by_side_gate_follows_the_size_limit_of_the_batched_chain(monkeypatch):
    """Up to the device's ``lml_batch`` limit the restarts of a fit of more than 128 points are stepped together as well
    (F6b: the reference's fit of 200 points), above it the thread farm / sequential loop takes over; d > 16 at N <= 128 has
    no single-launch objective and stays sequential."""
    g = load_golden("fit_mid")
    p = "f6b_k3_"
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "1")
    gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=4, random_state=3)
    gpr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=True)
    assert gpr.fit_stats.get("side_by_side") is True and gpr.n == 200
    assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_full"]) < 1e-5 * abs(float(g[p + "lml_full"]))
    np.testing.assert_allclose(gpr.kernel_.theta, g[p + "theta_full"], rtol=1e-3, atol=1e-3)
    assert gpr._can_step_restarts_together()
    monkeypatch.setattr(type(gpr.device), "lml_batch_max", 128)
    assert not gpr._can_step_restarts_together()
    monkeypatch.setattr(type(gpr.device), "lml_batch_max", 2048)
    monkeypatch.setattr(type(gpr), "n", property(lambda self: 100))
    monkeypatch.setattr(type(gpr), "d", property(lambda self: 20))
    assert not gpr._can_step_restarts_together()


def test_fit_contexts_are_dealt_out_over_the_devices_of_the_process(monkeypatch):
    """``fit_context_devices``: own device first, GPUs round-robin (the first restarts land on distinct
    GPUs), ``GPRY_HIP_FIT_CONTEXTS`` contexts per GPU, never more contexts than restarts; one rank of a
    multi-process launch stays on its own GPU; an explicit list (repeats allowed) is taken as it is."""
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "0")      # the thread farm is under test, not the side-by-side runs
    from gpry_amd import gpr as G
    from gpry_amd import _lib
    for var in ("GPRY_HIP_DEVICES", "WORLD_SIZE", "GPRY_HIP_FIT_CONTEXTS"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(_lib, "device_count", lambda: 8)
    assert G.fit_context_devices(0, 32) == [0, 1, 2, 3, 4, 5, 6, 7] * 3
    assert G.fit_context_devices(2, 5) == [2, 0, 1, 3, 4]
    assert G.fit_context_devices(0, 1) == [0]
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
    assert G.fit_context_devices(0, 32) == list(range(8))
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "2")
    monkeypatch.setenv("GPRY_HIP_DEVICES", "4,5")
    assert G.fit_context_devices(5, 32) == [5, 4, 5, 4]
    monkeypatch.setenv("GPRY_HIP_DEVICES", "none")
    assert G.fit_context_devices(3, 32) == [3, 3]
    monkeypatch.delenv("GPRY_HIP_DEVICES")
    for var in ("RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", "SLURM_STEP_ID", "SLURM_STEPID", "MPI_LOCALRANKID",
                "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_PROCID", "SLURM_STEP_NUM_TASKS", "SLURM_NTASKS"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "8")                 # one process per GPU: the ranks farm among themselves
    monkeypatch.setenv("RANK", "6")
    assert G.fit_context_devices(6, 32) == [6, 6]
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.delenv("RANK")
    # the reference's own parallel mode is mpi4py under mpirun / srun: no WORLD_SIZE there; a world size
    # WITHOUT the launcher's rank variable is a single process inside a multi-task allocation (`sbatch --ntasks=8` and no
    # srun): it keeps every GPU
    for var, rank_var in (("OMPI_COMM_WORLD_SIZE", "OMPI_COMM_WORLD_RANK"), ("PMI_SIZE", "PMI_RANK"), ("SLURM_STEP_NUM_TASKS", "SLURM_PROCID")):
        monkeypatch.setenv(var, "8")
        assert not G.multi_process_launch()
        assert G.fit_context_devices(6, 32)[:3] == [6, 0, 1]
        monkeypatch.setenv(rank_var, "0")
        assert G.multi_process_launch()
        assert G.fit_context_devices(6, 32) == [6, 6]
        monkeypatch.delenv(var)
        monkeypatch.delenv(rank_var)
    # ONE `python run.py` inside an interactive Slurm step of an 8-task allocation (salloc with
    # use_interactive_step, `srun --pty bash`) has a step id, a rank and the allocation's task count -- but a step of one
    # task: it is not a rank of many and keeps every GPU
    for k, v in (("SLURM_NTASKS", "8"), ("SLURM_STEP_ID", "0"), ("SLURM_PROCID", "0"), ("SLURM_STEP_NUM_TASKS", "1"), ("SLURM_LOCALID", "0")):
        monkeypatch.setenv(k, v)
    assert not G.multi_process_launch()
    assert G.fit_context_devices(6, 32)[:3] == [6, 0, 1]
    for k in ("SLURM_NTASKS", "SLURM_STEP_ID", "SLURM_PROCID", "SLURM_STEP_NUM_TASKS", "SLURM_LOCALID"):
        monkeypatch.delenv(k)
    monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_RANK", "5")
    assert G.default_device_index() == 5
    monkeypatch.delenv("OMPI_COMM_WORLD_LOCAL_RANK")
    assert G.fit_context_devices(0, 32, spec=[0, 0, 1, 1, 1]) == [0, 0, 1, 1, 1]
    monkeypatch.setattr(_lib, "device_count", lambda: 1)
    assert G.fit_context_devices(0, 32) == [0, 0]


def test_extra_fit_contexts_take_the_options_of_the_models_own_context():
    """contexts created for a fit (thread farm, side-by-side groups) evaluate with the options of the model's
    own context -- comparator factorisation, limits of the batched chain ... -- not with the defaults."""
    from gpry_amd import gpr as G

    class Ctx:
        def __init__(self, **kw):
            self.o = {k: 0 for k in G._FIT_CONTEXT_OPTIONS}
            self.o.update(kw)
            self.sets = []

        def get_option(self, k):
            return self.o[k]

        def set_option(self, k, v):
            self.o[k] = v
            self.sets.append(k)

    own, extra = Ctx(chol=1, lml_batch=2048, lml_batch_mb=512), Ctx(lml_batch=4096)
    G.copy_fit_options(own, extra)
    assert extra.o == own.o and sorted(extra.sets) == ["chol", "lml_batch", "lml_batch_mb"]
    G.copy_fit_options(own, object())          # a device double without options: nothing to do


@pytest.mark.parametrize("devices", [[0, 1], [0, 1, 2, 3, 4, 5, 6, 7], [0, 0, 1, 1, 2]])
def test_restart_farm_over_several_devices_in_one_process_equals_the_sequential_fit(monkeypatch, devices):
    """BASELINE configs[4] in ONE process (an unmodified ``Runner`` without mpi4py, gpry/run.py:315-325,
    1238-1293): the restarts of a fit spread over the contexts of ``fit_devices`` give the start points,
    optima, selected theta, LML and evaluation count of the reference's sequential loop, and every context
    is created on the device it was dealt."""
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "0")      # the thread farm is under test, not the side-by-side runs
    g = load_golden("fit")
    p = "f6_k3_"
    X, y = g[p + "X"], g[p + "y"]
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
    seq = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=9, random_state=11)
    seq.append_to_data(X[:60], y[:60], fit_gpr=True)
    assert not seq._fit_devs
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "3")
    par = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=9, random_state=11)
    par.fit_devices = devices
    par.append_to_data(X[:60], y[:60], fit_gpr=True)
    assert [par.device.device] + [idx for idx, _ in par._fit_devs] == devices
    assert all(dv.device == idx for idx, dv in par._fit_devs)
    assert par.fit_stats["devices"] == devices and sum(par.fit_stats["evals_per_context"]) > 0
    np.testing.assert_array_equal(par.kernel_.theta, seq.kernel_.theta)
    assert par.log_marginal_likelihood_value_ == seq.log_marginal_likelihood_value_
    assert par.n_eval_loglike == seq.n_eval_loglike
    np.testing.assert_array_equal(par.predict(g[p + "Xc"]), seq.predict(g[p + "Xc"]))
    # a second fit re-uses the contexts
    ids = [id(dv) for _, dv in par._fit_devs]
    par.append_to_data(X[60:], y[60:], fit_gpr=True)
    assert [id(dv) for _, dv in par._fit_devs] == ids


@pytest.mark.parametrize("devices,expect", [([0, 1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3]), ([0, 0, 0], [0, 0, 0]), ([0, 5], [0, 5])])
def test_side_by_side_groups_are_placed_on_the_devices_of_the_process(monkeypatch, devices, expect):
    """Several GPUs in ONE process: the independent groups of a side-by-side fit go where ``fit_context_devices`` deals the
    contexts of a fit (12 runs: at most 4 groups of >= 3 runs), every group's context on the device it was dealt; on one GPU
    ``batch_contexts`` groups share it.  The fit is the sequential fit bit for bit."""
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "1")
    g = load_golden("fit_mid")
    p = "f6b_k3_"
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
    monkeypatch.setenv("GPRY_HIP_FIT_BATCH_CONTEXTS", "1")
    seq = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=12, random_state=11)
    seq.fit_devices = [devices[0]]
    seq.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=True)
    assert seq.fit_stats["contexts"] == 1
    monkeypatch.setenv("GPRY_HIP_FIT_BATCH_CONTEXTS", "3")      # (what models of more than 300 points get by default)
    par = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=12, random_state=11)
    par.fit_devices = devices
    par.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=True)
    assert par.fit_stats["side_by_side"] and par.fit_stats["devices"] == expect
    assert [par.device.device] + [idx for idx, _ in par._fit_devs] == expect
    assert all(n > 0 for n in par.fit_stats["evals_per_context"])
    np.testing.assert_array_equal(par.kernel_.theta, seq.kernel_.theta)
    assert par.log_marginal_likelihood_value_ == seq.log_marginal_likelihood_value_
    assert par.n_eval_loglike == seq.n_eval_loglike


def test_host_append_fixed_theta_reproduces_reference_factor():
    """F8 through the host mirror: frozen theta, re-fitted pre-processors, lazy factor."""
    from gpry_amd.kernels import clone
    g = load_golden("predict")
    X, y, Xc = g["f8_X"], g["f8_y"], g["f8_Xc"]
    gpr = make_gpr(g["f8_bounds"], 2)
    k = clone(gpr.kernel)
    k.theta = g["f8_theta"]
    gpr.kernel_, gpr._fitted = k, True
    gpr.append_to_data(X[:32], y[:32], fit_gpr=False)
    np.testing.assert_allclose(gpr.predict_std(Xc), g["f8_std_before"], rtol=1e-7)
    gpr.append_to_data(X[32:], y[32:], fit_gpr=False, fit_classifier=False)
    np.testing.assert_allclose(gpr.L_, g["f8_L"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gpr.alpha_, g["f8_alpha_"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(gpr.predict(Xc), g["f8_mean_after"], rtol=1e-9, atol=1e-10)
    assert gpr.device.n_factorize == 2      # one factorisation per append, none per predict


def test_split_and_rank_generators():
    from gpry_amd.parallel import split_number_for_parallel_processes as split, get_random_generator
    assert list(split(5, 3)) == [2, 2, 1] and list(split(32, 8)) == [4] * 8
    assert list(split(2, 4)) == [1, 1, 0, 0] and list(split(0, 2)) == [0, 0]
    r = get_random_generator(7)
    ref = np.random.default_rng(np.random.SeedSequence(7).spawn(1)[0])
    assert r.integers(1 << 30) == ref.integers(1 << 30)
    assert get_random_generator(r) is r


def test_single_process_farm_equals_plain_fit():
    from gpry_amd.parallel import fit_gpr_parallel
    g = load_golden("fit")
    p = "f6_k3_"
    a = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=3, random_state=5)
    b = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=3, random_state=5)
    X, y = g[p + "X"][:40], g[p + "y"][:40]
    a.append_to_data(X, y, fit_gpr=True)
    lml, best, lmls = fit_gpr_parallel(b, X, y, comm=None, fit="full")
    assert best == 0 and lml == a.log_marginal_likelihood_value_
    np.testing.assert_array_equal(a.kernel_.theta, b.kernel_.theta)
    # fit=None keeps theta and only appends
    th = b.kernel_.theta.copy()
    fit_gpr_parallel(b, g[p + "X"][40:44], g[p + "y"][40:44], fit=None)
    np.testing.assert_array_equal(b.kernel_.theta, th)
    assert b.n == 44


# ---- two ranks over gloo -------------------------------------------------------------------
def _farm_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_multirank_cpu import GlooComm
        from gpry_amd.parallel import fit_gpr_parallel, get_random_generator
        g = load_golden("fit")
        p = "f6_k3_"
        comm = GlooComm()
        rng = get_random_generator(11, comm)
        gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=5, random_state=rng)
        X, y = g[p + "X"][:40], g[p + "y"][:40]
        lml, best, lmls = fit_gpr_parallel(gpr, X, y, comm=comm, fit="full")
        n_opt_lml = gpr.device.n_lml
        check = gpr.log_marginal_likelihood(gpr.kernel_.theta)
        m = gpr.predict(g[p + "Xc"][:16])
        # second round: 'simple' = one run per rank, then theta kept on a plain append
        lml2, best2, lmls2 = fit_gpr_parallel(gpr, g[p + "X"][40:50], g[p + "y"][40:50], comm=comm,
                                              fit="simple")
        th2 = gpr.kernel_.theta.copy()
        fit_gpr_parallel(gpr, g[p + "X"][50:52], g[p + "y"][50:52], comm=comm, fit=None)
        q.put((rank, lml, best, list(lmls), list(gpr.kernel_.theta), check, list(m), n_opt_lml,
               lml2, best2, list(th2), bool(np.array_equal(th2, gpr.kernel_.theta)), gpr.n))
    finally:
        dist.destroy_process_group()


def test_two_rank_restart_farm():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_farm_worker, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    r0, r1 = res
    # both ranks agree on the winner, hold its theta, and their factor reproduces its LML
    assert r0[1] == r1[1] == max(r0[3]) and r0[2] == r1[2] == int(np.argmax(r0[3]))
    assert r0[3] == r1[3] and r0[4] == r1[4]
    assert abs(r0[5] - r0[1]) < 1e-8 and abs(r1[5] - r1[1]) < 1e-8
    np.testing.assert_array_equal(r0[6], r1[6])
    assert r0[7] > 0 and r1[7] > 0          # 5 restarts -> 3 on rank 0, 2 on rank 1: both worked
    assert r0[8] == r1[8] and r0[9] == r1[9] and r0[10] == r1[10]
    assert r0[11] and r1[11] and r0[12] == r1[12] == 52

    # serial replay of the farm in this process: same starts, same optimiser, same winner
    g = load_golden("fit")
    p = "f6_k3_"
    X, y = g[p + "X"][:40], g[p + "y"][:40]
    lm = []
    for rank, n in ((0, 3), (1, 2)):
        rng = np.random.default_rng(np.random.SeedSequence(11).spawn(2)[rank])
        gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=5, random_state=rng)
        gpr.append_to_data(X, y, fit_gpr={"n_restarts": n, "start_from_current": rank == 0})
        lm.append(gpr.log_marginal_likelihood_value_)
    np.testing.assert_allclose(lm, r0[3], rtol=1e-12)


def test_int_seed_gives_each_rank_of_the_farm_its_own_start_points():
    """An int ``random_state`` must not make the ranks optimise duplicates: the farm hands rank r child r
    of ``SeedSequence(seed).spawn(world)`` (gpry/mpi.py:32-50, gpry/run.py:321,756)."""
    from gpry_amd.parallel import fit_gpr_parallel

    class FakeComm:
        world = 2

        def __init__(self, rank):
            self.rank = rank

        def allgather(self, a):
            return np.stack([np.asarray(a)] * self.world)

    g = load_golden("fit")
    p = "f6_k3_"
    X, y = g[p + "X"][:24], g[p + "y"][:24]
    draws = []
    for rank in range(2):
        gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=2, random_state=7)
        gpr.kernel_ = gpr.kernel.clone_with_theta(gpr.kernel.theta)
        fit_gpr_parallel(gpr, X, y, comm=FakeComm(rank), fit=None)
        assert isinstance(gpr.random_state, np.random.Generator)
        ref = np.random.default_rng(np.random.SeedSequence(7).spawn(2)[rank])
        draws.append(gpr.random_state.uniform(size=3))
        np.testing.assert_array_equal(draws[-1], ref.uniform(size=3))
    assert not np.array_equal(draws[0], draws[1])
    # a single process keeps the reference's single-rank behaviour (the seed is left alone)
    gpr = make_gpr(g[p + "bounds"], 3, random_state=7)
    gpr.kernel_ = gpr.kernel.clone_with_theta(gpr.kernel.theta)
    fit_gpr_parallel(gpr, X, y, comm=None, fit=None)
    assert gpr.random_state == 7


# ---- x-gradients through the host mirror (oracle arithmetic underneath) -----------------------
@pytest.mark.parametrize("kid", [0, 2, 3])
def test_host_predict_gradients_and_logexp_gradient_vs_reference(kid):
    """F10: tuple shapes, scalings (std_y once for the mean, twice for the std) and the LogExp
    gradient formula of the reference."""
    from gpry_amd.kernels import clone
    from gpry_amd.acquisition_functions import LogExp
    g = load_golden("gradients")
    p = f"f10_k{kid}_"
    gpr = make_gpr(g["f10_bounds"], kid)
    k = clone(gpr.kernel)
    k.theta = g[p + "theta"]
    gpr.kernel_, gpr._fitted = k, True
    gpr.append_to_data(g["f10_X"], g["f10_y"], fit_gpr=False)
    af = LogExp(dimension=3)
    for i, x in enumerate(g["f10_Xc"]):
        m, s, mg, sg = gpr.predict(x[None, :], return_std=True, return_mean_grad=True, return_std_grad=True)
        assert mg.shape == (3,) and sg.shape == (3,)
        np.testing.assert_allclose(m[0], g[p + "mean"][i], rtol=1e-9)
        np.testing.assert_allclose(mg, g[p + "mean_grad"][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sg, g[p + "std_grad"][i], rtol=1e-5, atol=1e-8)
        a, ag = af(x[None, :], gpr, eval_gradient=True)
        np.testing.assert_allclose(a[0], g[p + "acq"][i], rtol=1e-6, atol=1e-8)
        ref = g[p + "acq_grad"][i]
        assert np.array_equal(np.isinf(ag), np.isinf(ref))
        np.testing.assert_allclose(ag[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-5, atol=1e-7)
    m2, mg2 = gpr.predict(g["f10_Xc"][:1], return_mean_grad=True)
    np.testing.assert_allclose(mg2, g[p + "mean_grad"][0], rtol=1e-8, atol=1e-9)
    with pytest.raises(ValueError):
        gpr.predict(g["f10_Xc"][:2], return_std=True, return_mean_grad=True)
    with pytest.raises(ValueError):
        gpr.predict(g["f10_Xc"][:1], return_std_grad=True)


# ---- gates on the device -----------------------------------------
def _svm_problem(N=150, d=3, M=400, seed=9):
    from oracle import gpry_oracle as orc
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=seed)
    y = y.copy()
    y[X[:, 0] > 1.0] = -np.inf
    return bounds, X, y, Xc


def test_svm_device_params_reproduce_libsvm_verdicts():
    """The exported RBF expansion is the fitted SVC's decision function, and its sign is the
    classifier's verdict (gpry/svm.py:308-347)."""
    from gpry_amd.svm import SVM
    bounds, X, y, Xc = _svm_problem()
    svm = SVM(random_state=1)
    svm.fit(X, y, 20.0)
    sv, coef, gamma, intercept, pos = svm.device_params()
    dec = np.exp(-gamma * ((Xc[:, None, :] - sv[None, :, :]) ** 2).sum(-1)).dot(coef) + intercept
    np.testing.assert_allclose(dec, svm._svc.decision_function(Xc), rtol=1e-9, atol=1e-12)
    assert np.array_equal((dec > 0) == pos, svm.predict(Xc))
    assert 0 < svm.predict(Xc).sum() < len(Xc)
    empty = SVM()
    assert empty.device_params() is None
    allfin = SVM()
    allfin.fit(X[:20], np.arange(20.0), 1e9)
    assert allfin.all_finite and allfin.device_params() is None


def test_nora_with_device_gates_equals_host_masks():
    """Classifier + trust region evaluated by the (oracle-backed) device inside the sweep give the
    proposals, y and sigma that the host-side masks give."""
    from gpry_amd.kernels import clone
    from gpry_amd.gp_acquisition import NORA
    bounds, X, y, Xc = _svm_problem()
    res = []
    for use_device in (True, False):
        gpr = make_gpr(bounds, 3, account_for_inf="SVM", inf_threshold="20s", trust_region_factor=1.5,
                       random_state=1)
        k = clone(gpr.kernel)
        k.theta = np.log(np.array([4.0, 0.3, 0.3, 0.3]))
        gpr.kernel_, gpr._fitted = k, True
        gpr.append_to_data(X, y, fit_gpr=False)
        if not use_device:
            gpr._push_gates = lambda *a, **k: False
        acq = NORA(bounds, sampler="uniform", verbose=0)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
        Xp, yp, ap = acq.multi_add(gpr, n_points=3, bounds=gpr.trust_bounds, rng=np.random.default_rng(0))
        _, y_mc, s_mc, _ = acq.last_MC_sample()
        res.append((Xp, yp, ap, y_mc.copy(), s_mc.copy(), gpr.device))
    assert res[0][5].gates is not None and getattr(res[1][5], "gates", None) is None
    for a, b in zip(res[0][:5], res[1][:5]):
        np.testing.assert_array_equal(a, b)
    assert np.isneginf(res[0][3]).sum() > 0 and np.all(np.isfinite(res[0][1]))


def test_pool_arrays_are_fetched_from_the_device_only_on_demand():
    """One rank: NORA leaves y / sigma of the sweep on the device; ``last_MC_sample`` and a later
    reweighting fetch them (gpry/gp_acquisition.py:875-919 needs the old y)."""
    from gpry_amd.kernels import clone
    from gpry_amd.gp_acquisition import NORA
    from oracle import gpry_oracle as orc
    bounds, X, y, Xc = orc.synthetic_like_goldens(80, 3, 600, seed=5)
    gpr = make_gpr(bounds, 3)
    k = clone(gpr.kernel)
    k.theta = np.log(np.array([4.0, 0.3, 0.3, 0.3]))
    gpr.kernel_, gpr._fitted = k, True
    gpr.append_to_data(X, y, fit_gpr=False)
    acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0)
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    Xp, yp, ap = acq.multi_add(gpr, n_points=2, rng=np.random.default_rng(0))
    dev = gpr.device
    assert dev.n_fetch == 0 and acq._y_mc is None                 # nothing copied so far
    ref_y, ref_s = dev.y.copy(), dev.s.copy()
    Xs, ys, ss, ws = acq.last_MC_sample()
    assert dev.n_fetch == 1 and np.array_equal(ys, ref_y) and np.array_equal(ss, ref_s) and Xs is Xc
    acq.last_MC_sample()
    assert dev.n_fetch == 1                                         # cached after the first fetch
    # same thing when the first consumer is the reweighting of the next call
    acq2 = NORA(bounds, sampler="uniform", mc_every=2, verbose=0)
    acq2.do_MC_sample = acq.do_MC_sample
    acq2.multi_add(gpr, n_points=2, rng=np.random.default_rng(0))
    n0 = dev.n_fetch
    gpr.append_to_data(Xp, yp, fit_gpr=False)
    acq2.multi_add(gpr, n_points=2, rng=np.random.default_rng(0))   # reweights: needs the old y
    assert dev.n_fetch == n0 + 1 and acq2.is_last_MC_reweighted
    ref = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, gather_y=True)     # eager copies
    ref.do_MC_sample = acq.do_MC_sample
    gpr3 = make_gpr(bounds, 3)
    gpr3.kernel_, gpr3._fitted = clone(k), True
    gpr3.append_to_data(X, y, fit_gpr=False)
    a1 = ref.multi_add(gpr3, n_points=2, rng=np.random.default_rng(0))
    gpr3.append_to_data(a1[0], a1[1], fit_gpr=False)
    a2 = ref.multi_add(gpr3, n_points=2, rng=np.random.default_rng(0))
    np.testing.assert_array_equal(a1[0], Xp)
    np.testing.assert_array_equal(a2[0], acq2._X_already_proposed[-2:])


# ---- BatchOptimizer mirror (oracle arithmetic underneath) against the reference's run -----------------
def test_batch_optimizer_mirror_vs_reference_vectors():
    """F10b: same control flow, random-number order and optimiser as gpry/gp_acquisition.py:127-525; the
    L-BFGS-B runs see gradients that differ from the reference's in the last digits, so optima are
    compared to 1e-5 of the box."""
    from gpry_amd.gp_acquisition import BatchOptimizer
    from gpry_amd.proposal import UniformProposer
    from gpry_amd.kernels import clone
    g = load_golden("gradients")
    bounds = g["f10b_bounds"]

    def model():
        gpr = make_gpr(bounds, 3)
        k = clone(gpr.kernel)
        k.theta = g["f10b_theta"]
        gpr.kernel_, gpr._fitted = k, True
        gpr.append_to_data(g["f10b_X"], g["f10b_y"], fit_gpr=False)
        return gpr

    gpr = model()
    acq = BatchOptimizer(bounds, proposer=UniformProposer(bounds), n_restarts_optimizer=2, n_repeats_propose=1, verbose=0)
    rng = np.random.default_rng(9)
    x0, f0 = acq.optimize_acquisition_function(gpr, 0, bounds=bounds, rng=rng)
    x1, f1 = acq.optimize_acquisition_function(gpr, 1, bounds=bounds, rng=rng)
    np.testing.assert_allclose([x0, x1], g["f10b_opt_x"], atol=1e-4)
    np.testing.assert_allclose([float(f0), float(f1)], g["f10b_opt_f"], rtol=1e-6)
    gpr = model()
    acq = BatchOptimizer(bounds, proposer=UniformProposer(bounds), n_restarts_optimizer=3, n_repeats_propose=2, verbose=0)
    Xo, yl, av = acq.multi_add(gpr, n_points=3, rng=np.random.default_rng(5))
    np.testing.assert_allclose(Xo, g["f10b_X_opts"], atol=1e-4)
    np.testing.assert_allclose(yl, g["f10b_y_lies"], rtol=1e-6)
    np.testing.assert_allclose(av, g["f10b_acq_vals"], rtol=1e-6)
    assert gpr.n == 60                               # the lies went into a copy
    assert acq.stats["side_by_side"] is True         # the runs of a proposal advanced together, evaluations batched
    # ... and one after another (the reference's form): the same generator state afterwards, the same proposals
    gpr1, gpr2 = model(), model()
    kw = dict(proposer=None, n_restarts_optimizer=3, n_repeats_propose=2, verbose=0)
    r1, r2 = np.random.default_rng(5), np.random.default_rng(5)
    a1 = BatchOptimizer(bounds, lockstep=False, **kw)
    a2 = BatchOptimizer(bounds, **kw)
    o1 = a1.multi_add(gpr1, n_points=3, rng=r1)
    o2 = a2.multi_add(gpr2, n_points=3, rng=r2)
    assert a1.stats["side_by_side"] is False and a2.stats["side_by_side"] is True
    assert r1.random() == r2.random()
    np.testing.assert_allclose(o2[0], o1[0], atol=1e-5)
    np.testing.assert_allclose(o2[1], o1[1], rtol=1e-6)
    np.testing.assert_allclose(o2[2], o1[2], rtol=1e-6)
    with pytest.raises(TypeError):
        BatchOptimizer(bounds, proposer="uniform")
    with pytest.raises(ValueError):
        acq.multi_add(gpr, n_points=0)


def test_lockstep_driver_reproduces_fmin_l_bfgs_b():
    """``gpry_amd.lockstep``: scipy's L-BFGS-B routine driven for several starts at once takes, start by start, the steps of
    ``fmin_l_bfgs_b`` (same bits), whatever the mix of finished and running starts, with evaluations arriving in batches."""
    import scipy.optimize
    from gpry_amd import lockstep
    assert lockstep.available(), lockstep._STATE
    rng = np.random.default_rng(3)
    n = 5
    A = rng.standard_normal((n, n)); A = A @ A.T + 0.5 * np.eye(n)
    c = rng.standard_normal(n)

    def fg(x):
        return 0.5 * x @ A @ x - c @ x + np.sum(np.cos(2 * x)), A @ x - c - 2 * np.sin(2 * x)

    bnds = np.array([[-1.0, 1.0], [-0.3, 0.2], [-np.inf, 0.5], [-2.0, np.inf], [-np.inf, np.inf]])
    starts = rng.uniform(-1, 1, (9, n))
    batches = []

    def fgb(X):
        batches.append(len(X))
        out = [fg(x) for x in X]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    X, F, nfev = lockstep.minimize_lockstep(fgb, starts, bnds)
    for s0, x, f, ne in zip(starts, X, F, nfev):
        rx, rf, info = scipy.optimize.fmin_l_bfgs_b(fg, s0, bounds=[tuple(b) for b in bnds], approx_grad=False)
        np.testing.assert_array_equal(x, rx)
        assert f == rf and ne == info["funcalls"]
    assert batches[0] == 9 and min(batches) < 9 and sum(batches) == nfev.sum()


def _fresh_lockstep_state(monkeypatch):
    from gpry_amd import lockstep
    monkeypatch.setattr(lockstep, "_STATE", {"checked": False, "ok": False, "why": "", "warned": False, "how": ""})
    return lockstep


def test_lockstep_gate_goes_by_introspection_not_by_version_alone(monkeypatch):
    """``lockstep.available()``: the private routine is accepted when its argument list and the work arrays
    scipy's own wrapper allocates for it are the ones the driver uses, and a dry run behind canary padding reproduces
    ``fmin_l_bfgs_b`` -- whatever the version string says; it is refused, BEFORE the first call, when either differs."""
    import inspect
    import scipy
    from scipy.optimize import _lbfgsb, _lbfgsb_py
    # (a) a later release with the same routine: accepted (the version number alone used to refuse it)
    lockstep = _fresh_lockstep_state(monkeypatch)
    for ver, ok in (("1.14.1", False), ("1.16.0", False), ("2.0.0", False), ("1.15.0rc1", True), ("weird", False)):
        monkeypatch.setattr(scipy, "__version__", ver)
        assert lockstep._scipy_version_ok()[0] is ok
    monkeypatch.setattr(scipy, "__version__", "1.16.2")
    assert lockstep.available() and "1.16.2" in lockstep.how() and lockstep.why() == ""
    # (b) another argument list: refused without a call
    calls = []
    real = _lbfgsb.setulb

    def other_signature(*a, **k):
        calls.append(1)
    other_signature.__doc__ = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,iprint,csave,lsave,isave,dsave,maxls)"
    lockstep = _fresh_lockstep_state(monkeypatch)
    monkeypatch.setattr(_lbfgsb, "setulb", other_signature)
    assert not lockstep.available() and "argument list" in lockstep.why() and not calls
    monkeypatch.setattr(_lbfgsb, "setulb", real)
    # (c) scipy's wrapper allocates larger work arrays: refused without a call
    lockstep = _fresh_lockstep_state(monkeypatch)
    src = inspect.getsource(_lbfgsb_py._minimize_lbfgsb).replace("isave = zeros(44", "isave = zeros(48")
    monkeypatch.setattr(inspect, "getsource", lambda f: src)
    spy = []
    monkeypatch.setattr(_lbfgsb, "setulb", lambda *a, **k: spy.append(1))
    assert not lockstep.available() and "isave" in lockstep.why() and not spy
    monkeypatch.undo()


def test_lockstep_dry_run_catches_a_routine_that_writes_past_its_work_arrays(monkeypatch):
    """The self-check runs behind canary padding: a routine that writes beyond ``wa`` (larger work arrays than scipy
    1.15's) is caught by the dry run and refused."""
    from scipy.optimize import _lbfgsb
    lockstep = _fresh_lockstep_state(monkeypatch)
    real = _lbfgsb.setulb

    def overrunning(m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, *rest):
        if wa.base is not None:
            wa.base[wa.size + 3] = 1.0          # what a routine with a longer `wa` would do
        return real(m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, *rest)
    overrunning.__doc__ = real.__doc__
    monkeypatch.setattr(_lbfgsb, "setulb", overrunning)
    assert not lockstep.available() and "wrote past the end of `wa`" in lockstep.why()


def test_a_fit_that_cannot_run_side_by_side_says_so(monkeypatch):
    """When scipy's routine is refused, the multi-restart fit runs one restart after another, warns ONCE with the reason
    and records it: ``fit_stats["side_by_side"]`` is False and ``fit_stats["why"]`` carries the reason; the result is
    the side-by-side fit's."""
    import warnings
    from scipy.optimize import _lbfgsb
    g = load_golden("fit")
    p = "f6_k3_"
    X, y = g[p + "X"], g[p + "y"]
    monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "1")
    ref = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=4, random_state=7)
    ref.append_to_data(X[:60], y[:60], fit_gpr=True)
    assert ref.fit_stats["side_by_side"] and ref.fit_stats["why"] == ""
    lockstep = _fresh_lockstep_state(monkeypatch)
    real = _lbfgsb.setulb

    def fake(*a, **k):          # (scipy's own fmin_l_bfgs_b keeps working: only the docstring the gate reads differs)
        return real(*a, **k)
    fake.__doc__ = "setulb(some,other,arguments)"
    monkeypatch.setattr(_lbfgsb, "setulb", fake)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        gpr = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=4, random_state=7)
        gpr.append_to_data(X[:60], y[:60], fit_gpr=True)
        gpr2 = make_gpr(g[p + "bounds"], 3, n_restarts_optimizer=4, random_state=7)
        gpr2.append_to_data(X[:60], y[:60], fit_gpr=True)
    said = [w for w in rec if "one restart after another" in str(w.message)]
    assert len(said) == 1 and "argument list" in str(said[0].message)          # once per process, with the reason
    for m in (gpr, gpr2):
        assert m.fit_stats["side_by_side"] is False and "argument list" in m.fit_stats["why"]
    np.testing.assert_array_equal(gpr.kernel_.theta, ref.kernel_.theta)
    assert gpr.log_marginal_likelihood_value_ == ref.log_marginal_likelihood_value_


def test_proposers_draw_in_the_reference_order():
    from gpry_amd.proposal import UniformProposer, CentroidsProposer, PartialProposer
    import scipy.stats
    b = np.array([[-1.0, 2.0], [0.0, 4.0], [3.0, 5.0]])
    rng, ref = np.random.default_rng(4), np.random.default_rng(4)
    x = UniformProposer(b).get(rng)
    np.testing.assert_array_equal(x, scipy.stats.uniform(loc=b[:, 0], scale=b[:, 1] - b[:, 0]).rvs(size=3, random_state=ref))

    class G:
        X_train = np.random.default_rng(0).uniform(b[:, 0], b[:, 1], (12, 3))
    c = CentroidsProposer(b)
    c.update(G())
    c.update_bounds(b)
    # the centroids proposer against the reference's recipe written with scipy's frozen exponential
    rng, ref = np.random.default_rng(7), np.random.default_rng(7)
    sub = G.X_train[ref.choice(12, size=4, replace=False)]
    cen = np.average(sub, axis=0)
    partner = ref.choice(4, size=3, replace=False)
    kick = np.array([sub[j][i] for i, j in enumerate(partner)]) - cen
    kick *= scipy.stats.expon(scale=1.0).rvs(3, random_state=ref)
    np.testing.assert_array_equal(c.get(rng), np.clip(cen + kick, b[:, 0], b[:, 1]))
    assert rng.random() == ref.random()                      # and the generator was consumed alike
    p = PartialProposer(b, c)
    pts = np.array([p.get(np.random.default_rng(s)) for s in range(20)])
    assert np.all(pts >= b[:, 0]) and np.all(pts <= b[:, 1]) and len(np.unique(pts[:, 0])) > 10
    with pytest.raises(ValueError):
        PartialProposer(b, c, random_proposal_fraction=1.5)


def test_predict_with_gradients_and_border_path_of_the_mirror_on_the_cpu_double():
    """F10b through the host mirror with the oracle-backed double: the batch entry reproduces one reference call per
    point (scalings, zero-std rule), and a fixed-theta append with frozen pre-processors goes through
    ``append_rows`` while one that refits the pre-processors does not."""
    from gpry_amd.kernels import clone
    g = load_golden("gradients")
    gpr = make_gpr(g["f10b_bounds"], 3)
    k = clone(gpr.kernel)
    k.theta = g["f10b_theta"]
    gpr.kernel_, gpr._fitted = k, True
    gpr.append_to_data(g["f10b_X"][:50], g["f10b_y"][:50], fit_gpr=False)
    gpr.predict(g["f10b_Xq"][:1])                                   # brings the device factor up to date
    gpr.append_to_data(g["f10b_X"][50:55], g["f10b_y"][50:55], fit_gpr=False, fit_classifier=False)
    assert gpr.device.n_border == 1 and gpr.n_border_updates == 1 and gpr.device.N == 55
    gpr.append_to_data(g["f10b_X"][55:], g["f10b_y"][55:], fit_gpr=False)           # pre-processors refit: full path
    assert gpr.device.n_border == 1 and gpr.n == 60
    m, s, mg, sg = gpr.predict_with_gradients(g["f10b_Xq"])
    np.testing.assert_allclose(m, g["f10b_mean"], rtol=1e-9)
    np.testing.assert_allclose(s, g["f10b_std"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(mg, g["f10b_mean_grad"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sg, g["f10b_std_grad"], rtol=1e-4, atol=1e-6)


def test_predict_leaves_classifier_and_trust_box_to_a_device_that_applies_them():
    """``GaussianProcessRegressor.predict`` with GPry's defaults (account_for_inf="SVM", a trust region): when the
    context applies the gates itself (``applies_gates_in_predict``, the C option ``predict_gates``) the mirror sends no
    host mask and the gates travel to the device only when classifier, trust box or the ignore flag changed
    (``_sync_gates``); results equal the host-mask path (gpry/gpr.py:1107-1112, 1145-1150 -> gpry/svm.py:308-347)."""
    from oracle_device import OracleDevice
    from oracle import gpry_oracle as orc
    from gpry_amd.kernels import clone

    class GatedDouble(OracleDevice):
        applies_gates_in_predict = True
        n_set_gates = 0
        n_masked_calls = 0

        def set_gates(self, *a, **k):
            type(self).n_set_gates += 1
            super().set_gates(*a, **k)

        def predict(self, X, return_std=False, mask=None):
            if mask is not None:
                type(self).n_masked_calls += 1
            elif getattr(self, "gates", None) is not None:
                mask = self._gate_bits(np.atleast_2d(X))
            return super().predict(X, return_std=return_std, mask=mask)

        def predict_point(self, x, mask_bits=0, want_kinv=True):
            if mask_bits:
                type(self).n_masked_calls += 1
            return super().predict_point(x, mask_bits=mask_bits, want_kinv=want_kinv)

    bounds, X, y, Xc = orc.synthetic_like_goldens(150, 3, 400, seed=9)
    y = y.copy()
    y[X[:, 0] > 1.0] = -np.inf

    def build(kind):
        g = make_gpr(bounds, 3, account_for_inf="SVM", inf_threshold="20s", trust_region_factor=1.5, random_state=1)
        g._dev = kind()
        k = clone(g.kernel)
        k.theta = np.log(np.array([4.0, 0.3, 0.3, 0.3]))
        g.kernel_, g._fitted = k, True
        g.append_to_data(X, y, fit_gpr=False)
        return g

    dev_g, host_g = build(GatedDouble), build(OracleDevice)
    a = np.array([dev_g.predict(x[None, :], validate=False)[0] for x in Xc[:120]])
    b = np.array([host_g.predict(x[None, :], validate=False)[0] for x in Xc[:120]])
    np.testing.assert_array_equal(a, b)
    assert 5 < np.isneginf(a).sum() < 115
    assert GatedDouble.n_set_gates == 1 and GatedDouble.n_masked_calls == 0        # one upload for 120 calls, no host mask
    ma, sa = dev_g.predict(Xc, return_std=True)
    mb, sb = host_g.predict(Xc, return_std=True)
    np.testing.assert_array_equal(ma, mb)
    np.testing.assert_array_equal(sa, sb)
    assert GatedDouble.n_set_gates == 1
    # another gate set: the trust box switched off
    np.testing.assert_array_equal(dev_g.predict(Xc[:50], ignore_trust_region=True), host_g.predict(Xc[:50], ignore_trust_region=True))
    assert GatedDouble.n_set_gates == 2
    np.testing.assert_array_equal(dev_g.predict(Xc[:50]), host_g.predict(Xc[:50]))     # (same batch size: same BLAS sums)
    assert GatedDouble.n_set_gates == 3
    # the x-gradient branch reads the verdict: the one-point call applies the device's gates and returns it -- a finite
    # point, a classifier-rejected one and one outside the trust box, with the reference's conventions for each
    n1 = GatedDouble.n_set_gates
    rejected = np.flatnonzero(np.isneginf(a))
    for idx in (np.flatnonzero(np.isfinite(a))[0], rejected[0], rejected[-1]):
        x1 = Xc[idx][None, :]
        ga = dev_g.predict(x1, return_std=True, return_mean_grad=True, return_std_grad=True)
        gb = host_g.predict(x1, return_std=True, return_mean_grad=True, return_std_grad=True)
        for u, v in zip(ga, gb):
            np.testing.assert_array_equal(u, v)
    assert GatedDouble.n_masked_calls == 0 and dev_g.device.gates is not None and GatedDouble.n_set_gates == n1
    # the two-call entry points (without std) still take host verdicts: the device gates are cleared for them, and come back
    x1 = Xc[np.flatnonzero(np.isfinite(a))[0]][None, :]
    ga = dev_g.predict(x1, return_mean_grad=True)
    gb = host_g.predict(x1, return_mean_grad=True)
    for u, v in zip(ga, gb):
        np.testing.assert_array_equal(u, v)
    assert dev_g.device.gates is None
    np.testing.assert_array_equal(dev_g.predict(Xc[:50]), host_g.predict(Xc[:50]))
    assert dev_g.device.gates is not None
    # new data refits the classifier: its decision function reaches the device again
    n0 = GatedDouble.n_set_gates
    Xn = Xc[:20]
    yn = np.where(Xn[:, 0] > 0.5, -np.inf, -1.0)
    for g in (dev_g, host_g):
        g.append_to_data(Xn, yn, fit_gpr=False)
    np.testing.assert_array_equal(dev_g.predict(Xc[100:300]), host_g.predict(Xc[100:300]))
    assert GatedDouble.n_set_gates == n0 + 1


def test_schedule_of_a_side_by_side_fit(monkeypatch):
    """Round 6: the batched objective has two schedules (include/gpry_hip.h: "lml_schedule").  The host mirror takes the
    throughput one for the multi-restart fits (gpry/gpr.py:968-984, 10 + 2 d restarts by gpry/run.py:315-325) where whole fits
    were measured ahead -- from 2304 padded rows on, with at least six runs, all runs in one group --, the latency one (the bits
    of single evaluations) everywhere else; ``GPRY_HIP_FIT_SCHEDULE`` forces either."""
    from gpry_amd import gpr as G
    monkeypatch.delenv("GPRY_HIP_FIT_SCHEDULE", raising=False)
    monkeypatch.delenv("GPRY_HIP_FIT_TP_GROUPS", raising=False)
    assert G.fit_schedule(4096, 42) == ("throughput", 1)
    assert G.fit_schedule(2305, 6) == ("throughput", 1)          # (pads to 2432)
    assert G.fit_schedule(2304, 42) == ("throughput", 1)
    assert G.fit_schedule(2177, 42) == ("throughput", 1)         # (pads to 2304)
    assert G.fit_schedule(2176, 42) == ("latency", None)
    assert G.fit_schedule(1024, 26) == ("latency", None)
    assert G.fit_schedule(4096, 5) == ("latency", None)
    monkeypatch.setenv("GPRY_HIP_FIT_SCHEDULE", "latency")
    assert G.fit_schedule(4096, 42) == ("latency", None)
    monkeypatch.setenv("GPRY_HIP_FIT_SCHEDULE", "throughput")
    monkeypatch.setenv("GPRY_HIP_FIT_TP_GROUPS", "2")
    assert G.fit_schedule(300, 8) == ("throughput", 2)
    # the option travels to the extra contexts of a fit with the others
    assert {"lml_schedule", "lml_streams", "tp_block", "tp_tail"} <= set(G._FIT_CONTEXT_OPTIONS)
