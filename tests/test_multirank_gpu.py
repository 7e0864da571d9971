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
