"""Overhead of the in-process device group on ONE GPU: the configs[3] cycle (N=4096, d=16, M=1e6) with the
candidate pool sharded over k contexts of device 0.  The GPU is saturated by one context already, so
the cycle time should stay flat in k: what changes is the cost of sharding itself (k factorisations of
the replicated model, k partial last chunks, k top-k selections, the merge).
usage: python tools/bench_group.py [M] [k ...]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from gpry_amd.gp_acquisition import NORA

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
ks = [int(v) for v in sys.argv[2:]] or [1, 2, 4, 8]
N, d = 4096, 16
bounds, X, y, Xc, truth = bench.synthetic(N - d, d, M)
ref = None
for k in ks:
    gpr = bench.make_gpr(bounds)           # the same starting state for every k
    gpr.append_to_data(X, y, fit_gpr="simple")
    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0] * k)
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    rng = np.random.default_rng(2)
    Xn, _, _ = acq.multi_add(gpr, n_points=d, rng=rng)
    first = Xn.copy()
    t = []
    for it in range(4):
        bench.rewind(gpr, N - d)
        t0 = time.perf_counter()
        gpr.append_to_data(Xn, truth(Xn), fit_gpr="simple")
        t1 = time.perf_counter()
        Xn, _, _ = acq.multi_add(gpr, n_points=d, rng=rng)
        t.append((t1 - t0, time.perf_counter() - t1))
    t = np.array(t[1:])
    if ref is None:
        ref = (first, Xn.copy())
    print(json.dumps({"contexts": k, "refit_ms": t[:, 0].mean() * 1e3, "multi_add_ms": t[:, 1].mean() * 1e3,
                      "sweep_ms": acq.stats["sweep_s"] * 1e3, "rank_ms": acq.stats["rank_s"] * 1e3,
                      "shortlist": acq.stats["shortlist"], "same_proposals_as_k1": bool(np.array_equal(first, ref[0]) and np.array_equal(Xn, ref[1]))}))
    del acq, gpr
