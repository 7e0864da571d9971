"""world_size-2 test of the sharded NORA path on CPU (gloo): each rank sweeps its own
contiguous shard (the oracle stands in for the device), shortlists are exchanged with one
all-gather per round, and every rank must end with the single-rank reference pool."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GlooComm:
    """Test double with the interface of gpry_amd._lib.RcclComm, over torch.distributed/gloo."""

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.n_allgather = 0

    def allgather(self, arr):
        arr = np.ascontiguousarray(arr)
        t = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t)
        self.n_allgather += 1
        return np.stack([o.numpy().view(arr.dtype).reshape(arr.shape) for o in outs])


def _worker(rank, world, port, tag, shortlist, q, gather="auto"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_host_logic_cpu import FakeGPR, _golden_model
        from gpry_amd.gp_acquisition import NORA
        g, p, bounds, Xc, m = _golden_model(tag)
        gpr = FakeGPR(m)
        npts = len(g[p + "acq_cond"]) - 1
        comm = GlooComm()
        if gather is False:
            # nothing will need the pool's y / sigma (no reweighting): they are not gathered
            acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, comm=comm, shortlist_size=shortlist,
                       gather_y=False)
            acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
            Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
            ok = (np.array_equal(Xp, g[p + "X_pool"]) and np.allclose(ap, g[p + "acq_pool"], rtol=1e-8)
                  and acq.last_MC_sample()[1] is None and acq.last_MC_sample()[2] is None
                  and acq.last_MC_sample()[0] is Xc)
            q.put((rank, bool(ok), True, comm.n_allgather + 100, (acq._sweep_lo, acq._sweep_hi)))
            return
        acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, comm=comm, shortlist_size=shortlist)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
        Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        lo, hi = acq._sweep_lo, acq._sweep_hi
        ok = (np.array_equal(Xp, g[p + "X_pool"]) and np.allclose(ap, g[p + "acq_pool"], rtol=1e-8)
              and np.allclose(acq._y_mc, g[p + "y_mc"], rtol=1e-8, atol=1e-8)
              and len(gpr.device.acq) == hi - lo and hi - lo < len(Xc))
        gpr.append_to_data(Xp, g[p + "y_new"])
        Xp2, yp2, ap2 = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        ok2 = np.array_equal(Xp2, g[p + "X_pool2"]) and np.allclose(ap2, g[p + "acq_pool2"], rtol=1e-7)
        q.put((rank, bool(ok), bool(ok2), comm.n_allgather, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag,shortlist,gather,world", [("a", 8, "auto", 2), ("b", 64, "auto", 2),
                                                        ("b", 64, False, 2), ("a", 16, "auto", 3)])
def test_two_rank_sharded_multi_add_equals_single_rank_reference(tag, shortlist, gather, world):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tag, shortlist, q, gather)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    res.sort()
    shards = [r[4] for r in res]
    assert shards[0][0] == 0 and all(shards[i][1] == shards[i + 1][0] for i in range(world - 1))   # contiguous
    for rank, ok, ok2, n_ag, _ in res:
        assert ok, f"rank {rank}: first multi_add differs from the single-rank reference"
        assert ok2, f"rank {rank}: second (reweighted) multi_add differs"
        assert n_ag >= 4    # y/sigma gather + shortlist rounds, for both calls
