#!/usr/bin/env python3
"""Benchmark of the GP refit + NORA acquisition cycle (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One *step* is one active-learning cycle of GPry's hot path at configs[2] of
BASELINE.json (16-d synthetic posterior, N_train -> 4096, Matern-5/2, LogExp NORA sweep
over M = 1e6 candidates):

    gpr.append_to_data(X_new, y_new, fit_gpr="simple")   # L-BFGS-B refit, LML+grad on the GPU
    X_new, y_lie, acq = nora.multi_add(gpr, n_points=d)  # fused sweep + top-k + KB ranking

The training set grows by d points per step and reaches exactly N in the last timed
step (all steps share the padded size).  The candidate pool is resident in HBM before the
timed region.  With N > 1 GPUs (one process per GPU, launched by torch.distributed.run)
the refit is replicated and every rank sweeps its own M candidates of a pool of N*M
("weak"); ``--scaling strong`` shards a fixed pool of M instead.  Rank 0 prints ONE JSON
line; ``value`` = candidates swept by all ranks per second of whole cycle.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

# the host driver of the MI355X pool supports dmabuf IPC only (RCCL peer access over xGMI needs it);
# set before anything initialises HIP
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X public spec (FP64 matrix == FP64 vector); the guide's
#                               matrix-core table has no f64 row, see DESIGN.md
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synthetic(N, d, M, seed_train=0, seed_cand=1):
    """correlated-Gaussian log-posterior, candidates ~ N(0, 1.5 Sigma)."""
    rng = np.random.default_rng(seed_train)
    A = rng.standard_normal((d, d))
    Sigma = A @ A.T / d + 0.5 * np.eye(d)
    Lc = np.linalg.cholesky(Sigma)
    Sinv = np.linalg.inv(Sigma)
    bounds = np.array([[-5.0, 5.0]] * d)
    X = np.clip(rng.standard_normal((N, d)) @ Lc.T, -5, 5)

    def truth(P):
        return -0.5 * np.einsum("ni,ij,nj->n", P, Sinv, P)

    rng_c = np.random.default_rng(seed_cand)
    Xc = np.empty((M, d))
    for a in range(0, M, 1 << 18):   # chunked: keeps the host temporary small
        b = min(M, a + (1 << 18))
        Xc[a:b] = np.clip(rng_c.standard_normal((b - a, d)) @ (math.sqrt(1.5) * Lc).T, -5, 5)
    return bounds, X, truth(X), Xc, truth


def cpu_baseline(N, d, M, n_points, lml_evals, cache_models, budget_s=40.0):
    """Time the CPU oracle (numpy/scipy port of the reference path, same BLAS/LAPACK calls)
    on a bounded sample of the same workload and extrapolate one cycle."""
    from oracle import gpry_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        avail_gb = 16.0
    bounds, X, y, Xc, _ = synthetic(N, d, 4096)
    m = orc.OracleGPR(bounds, kernel_id=orc.MATERN52)
    m.theta = np.log(np.array([4.0] + [0.3] * d))
    m.fitted = True
    t0 = time.time()
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)   # kernel + cholesky + L^-1
    t_update = time.time() - t0
    t0 = time.time()
    m.predict(Xc, return_std=True)
    t_sweep = (time.time() - t0) / len(Xc)
    # LML+grad materialises (N, N, d+1) float64 tensors several times (~10 GB at N=4096)
    n_l = N if avail_gb > 24 else N // 2
    t0 = time.time()
    orc.log_marginal_likelihood(m.X_train_[:n_l], m.y_train_[:n_l], m.alpha[:n_l], m.theta,
                                orc.MATERN52, eval_gradient=True)
    t_lml = (time.time() - t0) * (N / n_l) ** 2    # O(N^2 d) tensors dominate: quadratic scaling
    cycle = lml_evals * t_lml + M * t_sweep + cache_models * t_update
    return {
        "value": M / cycle, "unit": "candidates/s", "cores": threads, "kind": "port",
        "sample": (f"oracle/gpry_oracle.py on host: 1 LML+grad at N={n_l} ({t_lml:.2f} s scaled to N={N}), "
                   f"predict+std on 4096 candidates ({t_sweep * 1e6:.1f} us each), 1 factor update "
                   f"({t_update:.2f} s); cycle = {lml_evals} LML evals + {M} candidates + {cache_models} "
                   f"conditioned-model refits = {cycle:.1f} s"),
        "cycle_s": cycle, "lml_grad_s": t_lml, "sweep_us_per_candidate": t_sweep * 1e6,
        "update_model_s": t_update,
    }


class _GlooComm:
    """Same interface as gpry_amd._lib.RcclComm over torch.distributed/gloo: used by the bench only
    if the RCCL communicator cannot be created (the exchanged shortlists are a few KB)."""

    def __init__(self, dist, dev):
        self._dist, self._dev = dist, dev
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def allgather(self, arr):
        import torch
        arr = np.ascontiguousarray(arr)
        t = torch.from_numpy(arr.view(np.uint8).reshape(-1).copy())
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t)
        return np.stack([o.numpy().view(arr.dtype).reshape(arr.shape) for o in outs])

    def allreduce_max(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return t.numpy()

    def barrier(self):
        self._dist.barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--M", type=int, default=1_000_000)
    ap.add_argument("--n-points", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--cpu-baseline", choices=["auto", "off"], default="auto")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node "
                  f"{args.gpus} bench.py ...`; running 1 GPU", file=sys.stderr)
        args.gpus = world
    os.environ["GPRY_HIP_DEVICE"] = str(local_rank)

    from gpry_amd import _lib
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.gp_acquisition import NORA
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y

    N, d, K, W = args.N, args.d, args.steps, args.warmup
    npts = args.n_points or d
    M_total = args.M * world if args.scaling == "weak" else args.M
    N0 = N - npts * (W + K)
    if N0 < 2 * d:
        raise SystemExit("N too small for the requested number of steps")
    bounds, X, y, Xc, truth = synthetic(N0, d, M_total)

    gpr = GaussianProcessRegressor(kernel={"Matern": {"nu": 2.5}}, bounds=bounds, noise_level=1e-2,
                                   preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
                                   account_for_inf=None, random_state=3, verbose=1)
    dev = gpr.device
    comm = None
    dist = None
    comm_kind = "none"
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        import torch.distributed as dist   # rendezvous only (CPU/gloo); the data path is RCCL
        dist.init_process_group("gloo")
        comm_kind = "rccl"
        box = [None]
        try:
            box = [_lib.RcclComm.unique_id() if rank == 0 else None]
        except Exception as e:
            print(f"bench.py[{rank}]: {e!r}", file=sys.stderr)
        dist.broadcast_object_list(box, src=0)
        # communicator + one verified exchange under a watchdog: a bootstrap that fails or hangs on
        # this node must not take the run down (the exchange is a few KB per cycle)
        made = {}

        def _connect():
            try:
                c = _lib.RcclComm(dev, world, rank, box[0])
                got = c.allgather(np.array([rank], dtype=np.int64))
                if list(got.ravel()) != list(range(world)):
                    raise RuntimeError(f"allgather self-test returned {got.ravel()}")
                made["comm"] = c
            except Exception as e:
                made["error"] = e

        ok = 0.0
        if box[0] is not None:
            import threading
            th = threading.Thread(target=_connect, daemon=True)
            th.start()
            th.join(timeout=float(os.environ.get("GPRY_BENCH_RCCL_TIMEOUT", "120")))
            if "comm" in made:
                comm, ok = made["comm"], 1.0
            else:
                why = repr(made.get("error", "timed out"))
                print(f"bench.py[{rank}]: RCCL communicator failed ({why}); using gloo for the "
                      f"shortlist exchange", file=sys.stderr)
        import torch
        flag = torch.tensor([ok])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # all ranks must agree on the transport
        if float(flag[0]) < 1.0:
            comm, comm_kind = _GlooComm(dist, dev), "gloo-fallback"

    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, comm=comm)
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    rng = np.random.default_rng(2)

    # ---- setup (untimed): first fit from the default init, first proposals, pool upload
    gpr.append_to_data(X, y, fit_gpr="simple")
    X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)

    host_t = {"refit": 0.0, "acq": 0.0}

    def step():
        nonlocal X_new
        t_a = time.perf_counter()
        gpr.append_to_data(X_new, truth(X_new), fit_gpr="simple")
        t_b = time.perf_counter()
        X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)
        host_t["refit"] += t_b - t_a
        host_t["acq"] += time.perf_counter() - t_b

    def fence():
        dev.sync()
        if comm is not None:
            comm.barrier()
        dev.sync()

    for _ in range(W):
        step()
    fence()
    dev.timing_reset()
    host_t["refit"] = host_t["acq"] = 0.0
    lml0 = gpr.n_eval_loglike
    cache_models = 0
    t0 = time.perf_counter()
    for _ in range(K):
        step()
        cache_models += acq.stats.get("cache_models", 0)
    fence()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
    lml_evals = (gpr.n_eval_loglike - lml0) / K
    assert gpr.n == N, (gpr.n, N)

    names = ("kernel_build", "potrf", "trtri", "lauum", "lml_traces", "cross_build", "sweep_gemm",
             "sweep_finish", "topk")
    T = {k: dev.timing(k) for k in names}
    per_step_ms = {k: T[k][0] / K for k in names}
    Np = (N + 127) // 128 * 128
    M_rank = acq._sweep_hi - acq._sweep_lo
    # dominant kernel: the FP64-MFMA triangular GEMM of the sweep.  Algorithmic flops per
    # candidate = N^2 (mul+add over the N^2/2 non-zeros of V) + 2N (mean), SURVEY.md 8(d)
    gemm_ms, gemm_n = T["sweep_gemm"]
    cand_per_launch = M_rank * K / max(gemm_n, 1)
    flops_launch = cand_per_launch * (float(N) ** 2 + 2.0 * N)
    achieved = flops_launch / (gemm_ms / max(gemm_n, 1) * 1e-3) / 1e12 if gemm_ms else 0.0
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        traffic = json.load(open(tfile)).get("sweep_gemm_hbm_bytes_per_launch")
    kb_ms, kb_n = T["kernel_build"]
    kb_gbps = (8.0 * N * N + 8.0 * N * d) / (kb_ms / max(kb_n, 1) * 1e-3) / 1e9 if kb_ms else 0.0
    sweep_ms = per_step_ms["cross_build"] + per_step_ms["sweep_gemm"] + per_step_ms["sweep_finish"]

    ms_per_step = elapsed / K * 1e3
    result = {
        "metric": "gp_refit_plus_nora_acq_cycle_throughput",
        "value": M_total / (elapsed / K),
        "unit": "candidates/s",
        "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: 16-d correlated-Gaussian posterior, N_train=4096, "
                               "Matern-5/2, LogExp NORA sweep M=1e6, n_points=16, fit_gpr='simple'",
                   "N_train": N, "d": d, "M_total": M_total, "M_per_gpu": M_rank, "n_points": npts,
                   "kernel": "ConstantKernel*Matern(nu=2.5)", "sharding": f"candidates x{world}",
                   "comm": comm_kind},
        "cycle": {"refit_plus_acq_ms": ms_per_step, "refit_ms": host_t["refit"] / K * 1e3,
                  "acquisition_ms": host_t["acq"] / K * 1e3,
                  "one_lml_grad_call_ms": (per_step_ms["kernel_build"] + per_step_ms["potrf"] + per_step_ms["trtri"] +
                                           per_step_ms["lauum"] + per_step_ms["lml_traces"]) / max(lml_evals, 1.0),
                  "lml_grad_evals_per_step": lml_evals,
                  "stage_ms_per_step": per_step_ms, "device_sweep_ms_per_step": sweep_ms,
                  "sweep_candidates_per_s_per_gpu": M_rank / (sweep_ms * 1e-3) if sweep_ms else None,
                  "shortlist": acq.stats.get("shortlist"), "cache_models_per_step": cache_models / K,
                  "rank_host_ms": acq.stats.get("rank_s", 0) * 1e3},
        "roofline": {"kernel": "sweep_gemm_dma_sp_kernel (V lower-triangular x K*^T panel, sum-of-squares epilogue)",
                     "bound": "mfma", "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "avg_launch_ms": gemm_ms / max(gemm_n, 1), "launches": gemm_n,
                     "flops_per_launch": flops_launch},
        "kernel_build": {"bound": "hbm", "achieved": kb_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": kb_gbps / HBM_PEAK_GBPS, "avg_launch_ms": kb_ms / max(kb_n, 1),
                         "bytes_per_launch": 8.0 * N * N + 8.0 * N * d},
    }
    # MFMA utilisation of the Cholesky (north_star): N^3/3 flop per factorisation over the average
    # duration of the `potrf` stage (64 fused panel steps + 15 MFMA trailing updates at N=4096), and of
    # the whole factor chain of one LML+gradient call (potrf + V = L^-1 + K^-1 = V^T V: N^3 flop)
    po_ms, po_n = T["potrf"]
    if po_n:
        tf = (float(Np) ** 3 / 3.0) / (po_ms / po_n * 1e-3) / 1e12
        chain_ms = sum(T[k][0] / max(T[k][1], 1) for k in ("potrf", "trtri", "lauum"))
        result["cholesky"] = {"bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": tf / F64_MFMA_PEAK_TFLOPS, "avg_ms": po_ms / po_n, "calls": po_n,
                              "flops_per_call": float(Np) ** 3 / 3.0,
                              "factor_chain_tflops": float(Np) ** 3 / (chain_ms * 1e-3) / 1e12}
    if rank == 0:
        # the build is one 40-us launch per LML evaluation: two events around a single launch also
        # time the dispatch gap, so quote the steady-state duration (50 launches back to back) too
        try:
            us = dev.microbench(6, 50)
            result["kernel_build"].update({
                "avg_launch_ms_back_to_back": us * 1e-3,
                "achieved_back_to_back": (8.0 * N * N + 8.0 * N * d) / (us * 1e-6) / 1e9,
                "frac_back_to_back": (8.0 * N * N + 8.0 * N * d) / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS})
        except Exception as e:
            result["kernel_build"]["back_to_back_error"] = repr(e)
        # measured ceilings of this very GPU, quoted beside the spec peaks used for `frac`
        try:
            result["measured_peaks"] = {
                "mfma_f64_vgpr_acc_TFLOPs": dev.microbench(2, 1),
                "hbm_copy_GBps": dev.microbench(1, 1 << 30),
                "hbm_fill_GBps": dev.microbench(3, 1 << 30)}
        except Exception as e:
            result["measured_peaks"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.cpu_baseline == "auto":
        try:
            result["cpu_baseline"] = cpu_baseline(N, d, args.M, npts, lml_evals, cache_models / K)
        except Exception as e:   # the baseline must never take the GPU number down with it
            result["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": None,
                                      "kind": "port", "sample": f"failed: {e!r}"}
    if comm is not None:
        comm.barrier()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
