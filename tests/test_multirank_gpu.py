"""One process per GPU over RCCL: runs only where at least two GPUs are visible (the development boxes
have one: a communicator cannot hold a device twice, so N > 1 over RCCL cannot be exercised there; the
identical host logic runs over gloo in tests/test_multirank_cpu.py and tests/test_fit_farm_cpu.py, the
sharding itself on one GPU in tests/test_group_gpu.py).  Each rank: its own device, an RcclComm built from
the id rank 0 publishes through a file, the sharded NORA.multi_add against the reference's F7 vectors, and
the restart farm's all-gather."""
import os
import sys
import tempfile
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    sys.path.insert(0, ROOT)
    from gpry_amd import _lib
    try:
        return _lib.device_count()
    except Exception:
        return 0


def _worker(rank, world, idfile, tag, q):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    os.environ["GPRY_HIP_DEVICE"] = str(rank)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from oracle import gpry_oracle as orc
    from gpry_amd import _lib
    from gpry_amd.gp_acquisition import NORA
    from test_host_mirror_gpu import make_gpr
    g = load_golden("multi_add")
    p = f"f7{tag}_"
    N, d = g[p + "X"].shape
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, int(g[p + "M"]), int(g[p + "seed"]))
    gpr = make_gpr(bounds, int(g[p + "kid"]), theta=g[p + "theta"])
    gpr.append_to_data(X, y, fit_gpr=False)
    if rank == 0:
        with open(idfile + ".tmp", "wb") as f:
            f.write(_lib.RcclComm.unique_id())
        os.replace(idfile + ".tmp", idfile)
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise RuntimeError("no RCCL id from rank 0")
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
    comm = _lib.RcclComm(gpr.device, world, rank, uid)
    try:
        got = comm.allgather(np.array([rank * 10 + 1], dtype=np.int64))
        ok_ag = list(got.ravel()) == [r * 10 + 1 for r in range(world)] and comm.info()[0] == world
        mx = comm.allreduce_max(np.array([float(rank), -float(rank)]))
        ok_ag = ok_ag and list(mx) == [world - 1.0, 0.0]
        npts = len(g[p + "acq_cond"]) - 1
        acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, comm=comm, shortlist_size=16)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
        Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        ok1 = np.array_equal(Xp, g[p + "X_pool"]) and np.allclose(ap, g[p + "acq_pool"], rtol=1e-7)
        gpr.append_to_data(Xp, g[p + "y_new"], fit_gpr=False)
        Xp2, yp2, ap2 = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        ok2 = np.array_equal(Xp2, g[p + "X_pool2"]) and np.allclose(ap2, g[p + "acq_pool2"], rtol=1e-6)
        q.put((rank, bool(ok_ag), bool(ok1), bool(ok2), (acq._sweep_lo, acq._sweep_hi)))
    finally:
        comm.close()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_rccl_ranks_shard_the_sweep_and_reproduce_the_reference(tag):
    world = min(_n_gpus(), 4)
    if world < 2:
        pytest.skip("needs at least two visible GPUs (RCCL cannot hold one device twice)")
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        idfile = os.path.join(tmp, "rccl_id")
        procs = [ctx.Process(target=_worker, args=(r, world, idfile, tag, q)) for r in range(world)]
        for pr in procs:
            pr.start()
        res = sorted(q.get(timeout=600) for _ in range(world))
        for pr in procs:
            pr.join(timeout=120)
            assert pr.exitcode == 0
    shards = [r[4] for r in res]
    assert shards[0][0] == 0 and all(shards[i][1] == shards[i + 1][0] for i in range(world - 1))
    for rank, ok_ag, ok1, ok2, _ in res:
        assert ok_ag, f"rank {rank}: RCCL all-gather / all-reduce self-test failed"
        assert ok1, f"rank {rank}: sharded multi_add differs from the reference"
        assert ok2, f"rank {rank}: reweighted second call differs"


@pytest.mark.parametrize("transport", ["rccl", "host"])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_device_group_over_distinct_gpus_reproduces_the_reference(tag, transport, monkeypatch):
    """The OTHER form of configs[3] (gpry/gp_acquisition.py:1148-1191 merged in one process): ONE process whose device group
    holds a context on each of the visible GPUs -- what an unmodified single-process ``gpry.Runner`` gets through
    ``NORA(devices=None)``.  With ``GPRY_GROUP_TRANSPORT=rccl`` the shortlist records travel by the in-process RCCL all-gather
    (``ncclCommInitAll``), which cannot be built on a 1-GPU box (a communicator holds a device once); with the default
    they are merged on the host.  Both must give the one-context proposals and the F7 vectors (both
    transports exercised the first time two GPUs are visible)."""
    n = min(_n_gpus(), 8)
    if n < 2:
        pytest.skip("needs at least two visible GPUs")
    monkeypatch.setenv("GPRY_GROUP_TRANSPORT", transport)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden
    from oracle import gpry_oracle as orc
    from gpry_amd.gp_acquisition import NORA
    from test_host_mirror_gpu import make_gpr
    g = load_golden("multi_add")
    p = f"f7{tag}_"
    N, d = g[p + "X"].shape
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, int(g[p + "M"]), int(g[p + "seed"]))
    gpr = make_gpr(bounds, int(g[p + "kid"]), theta=g[p + "theta"])
    gpr.append_to_data(X, y, fit_gpr=False)
    npts = len(g[p + "acq_cond"]) - 1
    acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, shortlist_size=8, devices=list(range(n)))
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    assert acq.stats["sweep_contexts"] == n
    assert acq._group is not None and acq._group.transport == transport, (acq._group.transport, acq._group.note)
    np.testing.assert_array_equal(Xp, g[p + "X_pool"])
    np.testing.assert_allclose(ap, g[p + "acq_pool"], rtol=1e-7)
    gpr.append_to_data(Xp, g[p + "y_new"], fit_gpr=False)
    Xp2, yp2, ap2 = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    np.testing.assert_array_equal(Xp2, g[p + "X_pool2"])
    np.testing.assert_allclose(ap2, g[p + "acq_pool2"], rtol=1e-6)


def test_restart_farm_over_distinct_gpus_in_one_process_equals_the_sequential_fit(monkeypatch):
    """configs[4] in ONE process on real devices: the restarts of a fit (thread farm: ``GPRY_HIP_FIT_LOCKSTEP=0``) are dealt
    out over a context on every visible GPU (``fit_context_devices``) and select what the sequential loop selects, bit for
    bit."""
    n = min(_n_gpus(), 8)
    if n < 2:
        pytest.skip("needs at least two visible GPUs")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import gpry_oracle as orc
    from test_host_mirror_gpu import make_gpr
    bounds, X, y, _ = orc.synthetic_like_goldens(2300, 6, 8, seed=5)
    monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "0")          # the thread farm is under test, not the side-by-side runs
    out = {}
    for mode in ("farm", "sequential"):
        monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
        gpr = make_gpr(bounds, 3, n_restarts_optimizer=max(4, n), random_state=5)
        gpr.fit_devices = list(range(n)) if mode == "farm" else [0]
        gpr.append_to_data(X, y, fit_gpr=True)
        out[mode] = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike, dict(getattr(gpr, "fit_stats", None) or {}))
    assert out["farm"][3].get("devices") == list(range(n))[:max(4, n)]
    np.testing.assert_array_equal(out["farm"][0], out["sequential"][0])
    assert out["farm"][1] == out["sequential"][1] and out["farm"][2] == out["sequential"][2]
