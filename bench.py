#!/usr/bin/env python3
"""Benchmark of the GP refit + NORA acquisition cycle (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload cycle|farm]

``--workload cycle`` (default; BASELINE configs[2] at N=1, configs[3] at N>1).  One *step* is one
active-learning cycle of GPry's hot path at EXACTLY N_train = 4096 (16-d synthetic posterior,
Matern-5/2, LogExp NORA sweep over M = 1e6 candidates):

    gpr.append_to_data(X_new, y_new, fit_gpr="simple")   # L-BFGS-B refit, LML+grad on the GPU
    X_new, y_lie, acq = nora.multi_add(gpr, n_points=d)  # fused sweep + top-k + KB ranking

Every step appends the d points proposed by the previous step to the SAME base of N - d training
rows (the base is restored by truncating the host arrays, microseconds, inside the timed region), so
every refit, factorisation and sweep of every timed step runs at N_train = N; ``config.
N_train_per_step`` lists what each step saw and the run aborts if one differs.  Every timed step gets a
FRESH candidate pool (two pre-generated pools in turn, SURVEY.md 8d / gpry/gp_acquisition.py:1023-1031): its
upload (128 MB) is inside the timed region, chunk by chunk underneath the sweep; ``cycle.
resident_pool_ms_per_step`` is the same cycle on a pool that stays in HBM (what rounds 1-4 reported as the
headline).  With N > 1 GPUs (one process per GPU, self-launched or by
torch.distributed.run) the refit is replicated and the SAME pool of M candidates is sharded N-way
(``--scaling strong``, the default: configs[3]); ``--scaling weak`` gives every rank M candidates.
The shortlists are exchanged over RCCL, the ranks meet over a small TCP store (``_TcpStore``; no torch on
that path); if the RCCL communicator cannot be built the run exits non-zero (``--allow-gloo`` permits the
gloo stand-in -- the only place ``torch`` is imported -- and says so in ``config.comm``).

``--workload farm`` (BASELINE configs[4]): N_train = 8192, d = 20; one step = one multi-restart
hyper-parameter fit, 32 L-BFGS-B restarts split over the ranks (gpry/run.py:1238-1293).

Rank 0 prints ONE JSON line; ``value`` = candidates swept by all ranks per second of whole cycle
(farm: restarts per second).

Launching.  ``python bench.py --gpus N`` with N > 1 and no ``WORLD_SIZE`` in the environment is its own
launcher: the parent (which imports neither the library nor HIP) starts N child processes of this script
with ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR=127.0.0.1`` / ``MASTER_PORT`` set -- the
environment ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` gives them --
relays rank 0's JSON line and exits with the worst child code; a watchdog ends the group when a rank dies
or the run exceeds ``GPRY_BENCH_LAUNCH_TIMEOUT`` seconds.  ``--gpus N`` beyond the visible GPUs is refused
(exit 2) unless ``GPRY_HIP_DEVICE_WRAP=1`` lets ranks share devices (development boxes; no RCCL then:
``--allow-gloo``).  ``--workload farm --mode group`` keeps ONE process and spreads the restarts of the fit
over N GPUs from host threads (what an unmodified single-process ``gpry.Runner`` does).
"""
import argparse
import json
import math
import os
import subprocess
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
CPU_CHUNK = 32768             # the CPU sweep is chunked at 32 768 candidates


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


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _cpu_leg(orc, N, d, n_sweep, lml_N, label):
    """One pass of the CPU port over a bounded sample: factor update, predict+std on ``n_sweep``
    candidates (one chunk of <= 32 768 rows), one LML+gradient at ``lml_N`` rows."""
    bounds, X, y, Xc, _ = synthetic(N, d, n_sweep)
    m = orc.OracleGPR(bounds, kernel_id=orc.MATERN52)
    m.theta = np.log(np.array([4.0] + [0.3] * d))
    m.fitted = True
    t0 = time.time()
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)   # kernel + cholesky + L^-1
    t_update = time.time() - t0
    t0 = time.time()
    m.predict(Xc, return_std=True)
    t_sweep = (time.time() - t0) / len(Xc)
    t0 = time.time()
    orc.log_marginal_likelihood(m.X_train_[:lml_N], m.y_train_[:lml_N], m.alpha[:lml_N], m.theta,
                                orc.MATERN52, eval_gradient=True)
    t_lml_raw = time.time() - t0
    t_lml = t_lml_raw * (N / lml_N) ** 2    # O(N^2 d) tensors dominate: quadratic scaling
    return {"label": label, "update_model_s": t_update, "sweep_us_per_candidate": t_sweep * 1e6,
            "sweep_sample": n_sweep, "lml_grad_s": t_lml, "lml_grad_measured_at_N": lml_N,
            "lml_grad_measured_s": t_lml_raw}


def cpu_baseline(N, d, M, lml_evals, cache_models):
    """Time the CPU oracle (numpy/scipy port of the reference path, same BLAS/LAPACK calls in the
    same order) on the box's host cores, on a bounded sample of the same workload, with all cores
    and with one thread, and extrapolate one cycle from each."""
    from oracle import gpry_oracle as orc
    from threadpoolctl import threadpool_info, threadpool_limits
    pools = threadpool_info()
    threads = max([p.get("num_threads", 1) for p in pools] or [1])
    blas = sorted({f"{p.get('internal_api')}-{p.get('version')}" for p in pools})
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        avail_gb = 16.0
    # LML+grad materialises (N, N, d+1) float64 tensors several times (~10 GB at N=4096)
    lml_N = N if avail_gb > 24 else N // 2
    legs = {}
    _cpu_leg(orc, 256, d, 256, 256, "warm-up (imports, thread pools)")
    legs["all_cores"] = _cpu_leg(orc, N, d, min(CPU_CHUNK, M), lml_N, f"{threads} BLAS threads")
    # one thread: same routines; the sample shrinks so that the leg stays within ~15 s
    lml_1 = lml_N if legs["all_cores"]["lml_grad_measured_s"] < 10.0 else lml_N // 2
    with threadpool_limits(limits=1):
        legs["single_thread"] = _cpu_leg(orc, N, d, min(2048, M), lml_1, "threadpool_limits(1) "
                                         "(= OMP_NUM_THREADS=1 for BLAS/LAPACK; numpy elementwise "
                                         "code is single-threaded in both legs)")
    out = {}
    for name, g in legs.items():
        cycle = lml_evals * g["lml_grad_s"] + M * g["sweep_us_per_candidate"] * 1e-6 + \
            cache_models * g["update_model_s"]
        g["cycle_s"] = cycle
        g["candidates_per_s"] = M / cycle
        out[name] = g
    a = out["all_cores"]
    # the small-N regime on the host: one-point mean predict and one LML + gradient of the CPU port (all BLAS threads)
    small = {}
    for Ns, ds in SMALL_N_CASES:
        bounds_s, Xs_, ys_, Xq_, theta_s = _small_problem(Ns, ds)
        m = orc.OracleGPR(bounds_s, kernel_id=orc.MATERN52)
        m.theta = theta_s
        m.fitted = True
        m.append_to_data(Xs_, ys_, fit_gpr=False, fit_preprocessors=True)
        small[f"N{Ns}_d{ds}"] = {
            "predict_1pt_us": _per_call_us(lambda: m.predict(Xq_[:1]), 300),
            "lml_grad_us": _per_call_us(lambda: orc.log_marginal_likelihood(m.X_train_, m.y_train_, m.alpha, theta_s, orc.MATERN52,
                                                                          eval_gradient=True), 20 if Ns <= 256 else 3)}
        if Ns == 256:
            # the full fit of the CPU port (same restarts, same optimiser), measured; at N = 1024 it is priced from the
            # evaluation count of the GPU fit x the per-evaluation time above (~1500 evaluations of ~0.1 s: minutes)
            mf = orc.OracleGPR(bounds_s, kernel_id=orc.MATERN52, n_restarts_optimizer=10 + 2 * ds, random_state=3)
            t0 = time.perf_counter()
            mf.append_to_data(Xs_, ys_, fit_gpr=True)
            small[f"N{Ns}_d{ds}"]["fit_full_s"] = time.perf_counter() - t0
            small[f"N{Ns}_d{ds}"]["fit_full_restarts"] = 10 + 2 * ds
    return {
        "small_n": small,
        "value": a["candidates_per_s"], "unit": "candidates/s", "cores": min(threads, affinity),
        "kind": "port",
        "sample": (f"oracle/gpry_oracle.py on host, {threads} BLAS threads: 1 LML+grad at N="
                   f"{a['lml_grad_measured_at_N']} ({a['lml_grad_measured_s']:.2f} s"
                   + ("" if a["lml_grad_measured_at_N"] == N else f", scaled x{(N / a['lml_grad_measured_at_N']) ** 2:.0f} to N={N}")
                   + f"), predict+std on one chunk of {a['sweep_sample']} candidates "
                   f"({a['sweep_us_per_candidate']:.1f} us each), 1 factor update ({a['update_model_s']:.2f} s); "
                   f"cycle = {lml_evals:g} LML evals + {M} candidates + {cache_models:g} conditioned-model "
                   f"refits = {a['cycle_s']:.1f} s"),
        "cpu_model": _cpu_model(), "nproc": os.cpu_count(), "affinity": affinity,
        "blas_threads": threads, "blas": blas, "chunk_rows": CPU_CHUNK,
        "cycle_s": a["cycle_s"], "lml_grad_s": a["lml_grad_s"],
        "sweep_us_per_candidate": a["sweep_us_per_candidate"], "update_model_s": a["update_model_s"],
        "single_thread": out["single_thread"],
    }


def measured_traffic(which, Np, M_launch_total):
    """HBM-side bytes per launch of the sweep contraction, from the PMC counters of THIS round's build
    (profiles/r06_traffic.json, written from tools/r06/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Returns (bytes or None, note): the figure is
    a measurement of the profiled run of that workload shape, not of this process; it is carried only when the shape of
    this run (padded training size, candidates per launch) is the profiled one."""
    path = os.path.join(ROOT, "profiles", "r06_traffic.json")
    if not os.path.exists(path):
        return None, "no PMC measurement of this round's build (profiles/r06_traffic.json)"
    rec = json.load(open(path)).get(which)
    if not rec:
        return None, f"profiles/r06_traffic.json has no entry {which!r}"
    if int(rec.get("Np", -1)) != int(Np):
        return None, f"measured at Np={rec.get('Np')}, this run has Np={Np}"
    note = (f"{rec['hbm_bytes_per_launch'] / 1e9:.2f} GB per launch of {rec['candidates_per_launch']} candidates = "
            f"{rec['hbm_bytes_per_launch'] / rec['algorithmic_bytes_per_launch']:.2f} x the {rec['algorithmic_bytes_per_launch'] / 1e9:.3f} GB "
            f"of operands ({rec['source']})")
    return rec["hbm_bytes_per_launch"], note


class _GlooComm:
    """Same interface as gpry_amd._lib.RcclComm over torch.distributed/gloo: used by the bench only
    with ``--allow-gloo`` when the RCCL communicator cannot be created."""

    def __init__(self, dist):
        self._dist = dist
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


class _TcpStore:
    """The rendezvous of an N-GPU run without torch: rank 0 listens next to ``MASTER_PORT`` on ``MASTER_ADDR``, every other
    rank keeps ONE connection to it.  It carries the 128-byte RCCL id, the ranks' agreement on the transport and the
    closing barriers -- a few hundred bytes per run; the shortlists travel over RCCL.  (Under ``torch.distributed.run``
    the agent's own store sits on ``MASTER_PORT`` itself, hence the offset; a handshake word tells a rank that it has
    reached this store and not some other service on a neighbouring port.)"""
    MAGIC = b"gpry-bench-store-1"
    PORT_OFFSETS = (1, 2, 3, 5, 8, 13, 21, 34)

    def __init__(self, rank, world, addr=None, port=None, timeout=None):
        import socket
        import struct
        self.rank, self.world, self._struct = rank, world, struct
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port if port is not None else os.environ.get("MASTER_PORT", "29500"))
        timeout = float(timeout if timeout is not None else os.environ.get("GPRY_BENCH_STORE_TIMEOUT", "300"))
        self._peers = {}
        deadline = time.monotonic() + timeout
        if rank == 0:
            srv = None
            for off in self.PORT_OFFSETS:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port + off))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError(f"bench store: no free port next to {addr}:{port}")
            srv.listen(world)
            srv.settimeout(1.0)
            while len(self._peers) < world - 1:
                if time.monotonic() > deadline:
                    raise RuntimeError(f"bench store: only {len(self._peers) + 1} of {world} ranks arrived within {timeout:.0f} s")
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.settimeout(timeout)
                hello = self._recv_from(conn)
                if not hello.startswith(self.MAGIC):
                    conn.close()
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self._peers[int(hello[len(self.MAGIC):])] = conn
                self._send_to(conn, self.MAGIC)
            srv.close()
        else:
            sock = None
            while sock is None:
                for off in self.PORT_OFFSETS:
                    try:
                        c = socket.create_connection((addr, port + off), timeout=2.0)
                        c.settimeout(10.0)
                        self._send_to(c, self.MAGIC + str(rank).encode())
                        if self._recv_from(c) == self.MAGIC:
                            sock = c
                            break
                        c.close()
                    except OSError:
                        pass
                if sock is None:
                    if time.monotonic() > deadline:
                        raise RuntimeError(f"bench store: rank {rank} found no store next to {addr}:{port} within {timeout:.0f} s")
                    time.sleep(0.2)
            sock.settimeout(timeout)
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._peers[0] = sock

    def _send_to(self, sock, payload):
        sock.sendall(self._struct.pack("<I", len(payload)) + payload)

    def _recv_from(self, sock):
        def exactly(n):
            buf = b""
            while len(buf) < n:
                part = sock.recv(n - len(buf))
                if not part:
                    raise RuntimeError("bench store: a peer closed its connection")
                buf += part
            return buf
        return exactly(self._struct.unpack("<I", exactly(4))[0])

    def broadcast(self, payload):
        """bytes of rank 0 to every rank"""
        if self.rank == 0:
            for r in sorted(self._peers):
                self._send_to(self._peers[r], payload)
            return payload
        return self._recv_from(self._peers[0])

    def all_min(self, value):
        """the smallest of the ranks' numbers, on every rank (gather at rank 0, answer to all)"""
        if self.rank == 0:
            vals = [float(value)] + [float(self._recv_from(self._peers[r]).decode()) for r in sorted(self._peers)]
            return float(self.broadcast(repr(min(vals)).encode()).decode())
        self._send_to(self._peers[0], repr(float(value)).encode())
        return float(self.broadcast(b"").decode())

    def barrier(self):
        self.all_min(0.0)

    def destroy_process_group(self):      # (the name the closing code calls on either kind of rendezvous object)
        for c in self._peers.values():
            try:
                c.close()
            except OSError:
                pass
        self._peers = {}


def connect(args, rank, world, dev):
    """Rendezvous over a small TCP store (no torch), data-path communicator over RCCL.  Returns (comm, rendezvous, kind,
    rccl_ranks).  A failed or hanging RCCL bootstrap ends the run with a non-zero exit code unless ``--allow-gloo`` was
    given -- only then is ``torch.distributed`` imported, for its gloo backend; the process never continues with a thread
    stuck inside RCCL."""
    from gpry_amd import _lib
    store = _TcpStore(rank, world)
    uid, err = b"", None
    if rank == 0:
        try:
            uid = bytes(_lib.RcclComm.unique_id())
        except Exception as e:
            err = repr(e)
    uid = store.broadcast(uid)
    made = {}

    def _connect():
        try:
            c = _lib.RcclComm(dev, world, rank, uid)
            got = c.allgather(np.array([rank], dtype=np.int64))
            if list(got.ravel()) != list(range(world)):
                raise RuntimeError(f"allgather self-test returned {got.ravel()}")
            made["comm"] = c
        except Exception as e:
            made["error"] = repr(e)

    if uid:
        import threading
        th = threading.Thread(target=_connect, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("GPRY_BENCH_RCCL_TIMEOUT", "180")))
        if th.is_alive():
            print(f"bench.py[{rank}]: RCCL bootstrap still hanging after the time limit; giving up",
                  file=sys.stderr, flush=True)
            os._exit(4)      # a thread is stuck inside ncclCommInitRank: nothing sane can follow
        err = made.get("error")
    if store.all_min(1.0 if "comm" in made else 0.0) >= 1.0:          # all ranks must agree on the transport
        n_rccl = made["comm"].info()[0]
        return made["comm"], store, "rccl", n_rccl
    if "comm" in made:
        made["comm"].close()
    msg = f"bench.py[{rank}]: no RCCL communicator over {world} ranks ({err or 'a peer failed'})"
    if world > 1 and not args.allow_gloo:
        print(msg + "; exiting (pass --allow-gloo to exchange the shortlists over gloo instead)",
              file=sys.stderr, flush=True)
        store.barrier()
        store.destroy_process_group()
        sys.exit(3)
    print(msg + "; using gloo (--allow-gloo)", file=sys.stderr, flush=True)
    store.barrier()
    store.destroy_process_group()
    import torch.distributed as dist   # --allow-gloo only: the shortlists over gloo (development boxes without RCCL)
    dist.init_process_group("gloo")
    return _GlooComm(dist), dist, "gloo-fallback", 0


def rewind(gpr, n_base):
    """Bench-only: drop the rows appended by the previous step so that the next ``append_to_data``
    arrives at exactly N again.  ``X_train_all`` / ``y_train_all`` are public attributes
    (gpry/gpr.py:283); everything else is recomputed from them by ``append_to_data``."""
    gpr.X_train_all = gpr.X_train_all[:n_base]
    gpr.y_train_all = gpr.y_train_all[:n_base]


def make_gpr(bounds, **kw):
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
    kw.setdefault("random_state", 3)
    return GaussianProcessRegressor(kernel={"Matern": {"nu": 2.5}}, bounds=bounds, noise_level=1e-2,
                                    preprocessing_X=Normalize_bounds(bounds), preprocessing_y=Normalize_y(),
                                    account_for_inf=None, verbose=1, **kw)


def refit_extras(bounds, X, y):
    """Not the headline: the hyper-parameter fit away from a converged theta, at the same N.
    (a) ``fit_gpr='simple'`` from the default init C=10, l=0.1 (gpry/gpr.py:351-352) -- the ~60
    evaluation case BASELINE.md section 2 prices at 32 min on the CPU; (b) a 4-restart full fit."""
    from gpry_amd.kernels import clone
    out = {}
    d = X.shape[1]
    # (c) round 6: the fit the reference's Runner asks for every fit_full_every-th iteration -- n_restarts = 10 + 2 d
    # (gpry/run.py:315-325), first run from the current theta (gpry/gpr.py:968-984) -- at the headline size
    for name, fit in (("cold_simple_from_default_init", "simple"),
                      ("full_fit_4_restarts", {"n_restarts": 4, "start_from_current": True}),
                      ("full_fit_default_restarts", {"n_restarts": 10 + 2 * d, "start_from_current": True})):
        g = make_gpr(bounds)
        g.kernel_ = clone(g.kernel)
        g._fitted = True                      # 'simple' then starts at kernel_.theta = the default init
        g.device.set_train(np.zeros((len(X), X.shape[1])), np.zeros(len(X)), 1e-4)   # allocation: untimed
        e0 = g.n_eval_loglike
        t0 = time.perf_counter()
        g.append_to_data(X, y, fit_gpr=fit)
        g.device.sync()
        dt = time.perf_counter() - t0
        ev = g.n_eval_loglike - e0
        stats = getattr(g, "fit_stats", None) or {}
        out[name] = {"ms": dt * 1e3, "lml_grad_evals": ev, "ms_per_eval": dt * 1e3 / max(ev, 1),
                     "lml": float(g.log_marginal_likelihood_value_), "N_train": g.n,
                     "n_restarts": 1 if fit == "simple" else fit["n_restarts"],
                     "schedule": stats.get("schedule", "latency"), "groups": stats.get("contexts", 1),
                     "side_by_side": bool(stats.get("side_by_side")),
                     "rounds": int(max(stats.get("evals_per_run", [0]) or [0]))}
        if name == "full_fit_default_restarts":
            # the same fit once more on the warm contexts (scratch arenas, plans): what a run pays from its second full fit on
            g.kernel_ = clone(g.kernel)
            g.set_random_state(3)
            e0 = g.n_eval_loglike
            t0 = time.perf_counter()
            g.newly_appended_for_inv = 1
            g.fit_gpr_hyperparameters(n_restarts=fit["n_restarts"], start_from_current=True)
            g.device.sync()
            out[name].update({"ms_warm": (time.perf_counter() - t0) * 1e3, "lml_grad_evals_warm": g.n_eval_loglike - e0})
        del g
    return out


SMALL_N_CASES = ((64, 2), (128, 4), (256, 4), (1024, 8))


def _small_problem(N, d):
    rng = np.random.default_rng(100 + N)
    bounds = np.array([[-5.0, 5.0]] * d)
    X = rng.uniform(-5, 5, size=(N, d))
    y = -0.5 * (X ** 2).sum(axis=1) / d
    return bounds, X, y, rng.uniform(-5, 5, size=(512, d)), np.log(np.array([4.0] + [0.3] * d))


def _per_call_us(fn, reps):
    for _ in range(max(3, reps // 10)):
        fn()
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    return best


def small_n_extras():
    """Not the headline: the regime most GPry iterations live in: one-point
    mean ``predict`` -- what PolyChord / UltraNest / MCMC call 1e4..1e6 times per acquisition step,
    gpry/gp_acquisition.py:766-771 -- and one LML + gradient evaluation at small N, through the mirror class, wall clock per
    call; the resident predict kernel (csrc/server.hip) and the single-launch objective (csrc/lml_small.hip) against the
    one-launch-per-call / 19-kernel paths of round 2."""
    from gpry_amd.kernels import clone
    out = {}
    for N, d in SMALL_N_CASES:
        bounds, X, y, Xq, theta = _small_problem(N, d)
        g = make_gpr(bounds)
        k = clone(g.kernel)
        k.theta = theta
        g.kernel_, g._fitted = k, True
        g.append_to_data(X, y, fit_gpr=False)
        dev = g.device
        it = {"i": 0}

        def one_point():
            it["i"] = (it["i"] + 1) % len(Xq)
            return g.predict(Xq[it["i"]:it["i"] + 1], return_std=False, validate=False)

        row = {}
        for serve in (1, 0):
            dev.set_option("predict_serve", serve)
            row["predict_1pt_us" if serve else "predict_1pt_us_one_launch_per_call"] = _per_call_us(one_point, 2000)
        dev.set_option("predict_serve", 1)
        for small in (1, 0):
            dev.set_option("lml_small", small)
            key = "lml_grad_us" if small else "lml_grad_us_general_chain"
            row[key] = _per_call_us(lambda: g.log_marginal_likelihood(theta, eval_gradient=True), 200 if N <= 256 else 50)
        dev.set_option("lml_small", 1)
        row["single_launch_objective"] = bool(N <= 128 and d <= 16)
        if N >= 256:
            # the fit these sizes are refitted with (gpry/run.py:315-325: 10 + 2 d restarts, gpry/gpr.py:968-984 one after
            # another there): restarts stepped side by side, a round's thetas through ONE chain of launches (gpry_lml_batch)
            thetas = theta + np.random.default_rng(5).uniform(-0.3, 0.3, (32, d + 1))
            full = np.array([np.concatenate(([t[0]], t[1:])) for t in thetas])
            row["lml_grad_batch32_ms"] = _per_call_us(lambda: dev.lml_batch(full, True), 10) * 1e-3
            try:        # the throughput schedule of the same call (option "lml_schedule" = 1, round 6)
                dev.set_option("lml_schedule", 1)
                row["lml_grad_batch32_throughput_ms"] = _per_call_us(lambda: dev.lml_batch(full, True), 10) * 1e-3
            finally:
                dev.set_option("lml_schedule", 0)
            # ONE model refitted, as in a run (gpry/run.py keeps its regressor): contexts and scratch arenas persist; the
            # first fit, which creates them, is not the one timed
            g2 = make_gpr(bounds, n_restarts_optimizer=10 + 2 * d)
            g2.append_to_data(X, y, fit_gpr=False)
            best, nev = None, 0
            for rep in range(3):
                g2.set_random_state(3)
                e0 = g2.n_eval_loglike
                # (as behind append_to_data: the closing _update_model of the fit factorises -- without this it returned
                # early with "No new points have been appended", and the figure left the factorisation out)
                g2.newly_appended_for_inv = 1
                t0 = time.perf_counter()
                g2.fit_gpr_hyperparameters(start_from_current=False)
                dt = time.perf_counter() - t0
                nev = g2.n_eval_loglike - e0
                if rep:
                    best = dt if best is None else min(best, dt)
            stats = getattr(g2, "fit_stats", None) or {}
            row.update({"fit_full_ms": best * 1e3, "fit_full_restarts": 10 + 2 * d, "fit_full_evals": int(nev),
                        "fit_full_groups": int(stats.get("contexts", 1)), "fit_full_schedule": stats.get("schedule", "latency"),
                        "fit_full_side_by_side": bool(stats.get("side_by_side")),
                        "fit_full_side_by_side_why_not": stats.get("why", ""),
                        "fit_full_rounds": int(max(stats.get("evals_per_run", [0])))})
            del g2
        out[f"N{N}_d{d}"] = row
        del g
    # which scipy drove the fits and whether its private L-BFGS-B routine passed the side-by-side gate (gpry_amd/lockstep.py)
    import scipy
    from gpry_amd import lockstep
    out["scipy"] = {"version": scipy.__version__, "side_by_side_available": bool(lockstep.available()),
                    "accepted_on": lockstep.how(), "refused_because": lockstep.why()}
    return out


def run_farm(args, rank, world, local_rank):
    """BASELINE configs[4]: N=8192, d=20, 32 L-BFGS-B restarts of the LML farmed over the GPUs."""
    from gpry_amd.parallel import fit_gpr_parallel, split_number_for_parallel_processes
    N, d, K, W = args.N or 8192, args.d or 20, args.steps, args.warmup
    n_restarts = args.restarts
    bounds, X, y, _, truth = synthetic(N, d, 16)
    n_base = N - d
    gpr = make_gpr(bounds, n_restarts_optimizer=n_restarts)
    dev = gpr.device
    comm, dist, comm_kind, n_rccl = None, None, "none", 0
    group = args.mode == "group"
    n_gpus = args.gpus if group else world
    if group:
        # ONE process: the contexts of the fit are dealt out over the GPUs (gpry_amd.gpr.fit_context_devices;
        # an explicit list here so that a development box can repeat its only device)
        from gpry_amd.gpr import fit_contexts
        devs = [int(v) for v in os.environ.get("GPRY_BENCH_GROUP_DEVICES", "0").split(",")][:args.gpus]
        gpr.fit_devices = [devs[i % len(devs)] for i in range(fit_contexts() * len(devs))]
        comm_kind = "one process, host threads"
    elif world > 1 or "RANK" in os.environ:
        comm, dist, comm_kind, n_rccl = connect(args, rank, world, dev)
    # setup (untimed): first fit on the base set
    gpr.append_to_data(X[:n_base], y[:n_base], fit_gpr="simple")
    X_new, y_new = X[n_base:], y[n_base:]

    def step():
        rewind(gpr, n_base)
        fit_gpr_parallel(gpr, X_new, y_new, comm=comm, fit="full", n_restarts=n_restarts)
        assert gpr.n == N, (gpr.n, N)

    def fence():
        dev.sync()
        if comm is not None:
            comm.barrier()
        dev.sync()

    for _ in range(W):
        step()
    fence()
    dev.timing_reset()
    e0 = gpr.n_eval_loglike
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    evals_rank = (gpr.n_eval_loglike - e0) / K
    if comm is not None:
        elapsed = float(comm.allreduce_max(np.array([elapsed]))[0])
        evals_all = comm.allgather(np.array([evals_rank]))[:, 0]
    else:
        evals_all = np.array([evals_rank])
    names = ("kernel_build", "potrf", "trtri", "lauum", "lml_traces")
    T = {k: dev.timing(k) for k in names}
    Np = (N + 127) // 128 * 128
    evals_gpu = evals_rank / (n_gpus if group else 1)      # group: the process' evaluations are spread over its GPUs
    one_eval_ms = sum(T[k][0] / max(T[k][1], 1) for k in names)
    po_ms, po_n = T["potrf"]
    chain_ms = sum(T[k][0] / max(T[k][1], 1) for k in ("potrf", "trtri", "lauum"))
    result = {
        "metric": "gp_hyperparameter_restart_farm_throughput",
        "value": n_restarts / (elapsed / K), "unit": "restarts/s",
        "n_gpus": n_gpus, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: 20-d synthetic posterior, N_train=8192, Matern-5/2, "
                               f"{n_restarts} L-BFGS-B restarts of the log-marginal-likelihood " +
                               ("spread over the GPUs by ONE process (host threads, one context each)" if group
                                else "split over the GPUs (one process per GPU)"),
                   "N_train": N, "d": d, "n_restarts": n_restarts, "mode": args.mode,
                   "restarts_per_rank": [int(v) for v in split_number_for_parallel_processes(n_restarts, world)],
                   "fit_context_devices": getattr(gpr, "fit_stats", {}).get("devices"),
                   "evals_per_context_last_fit": getattr(gpr, "fit_stats", {}).get("evals_per_context"),
                   "comm": comm_kind, "rccl_ranks": n_rccl},
        "farm": {"lml_grad_evals_per_step_per_rank": [float(v) for v in evals_all],
                 "one_lml_grad_call_ms_device": one_eval_ms,
                 "stage_ms_per_eval": {k: T[k][0] / max(T[k][1], 1) for k in names}},
        # the restarts of a rank are worked off by up to three device contexts at once (DESIGN.md
        # section 5), so the per-stage event times above overlap; the rate of the GPU as a whole is
        # evaluations x Np^3 flop (potrf + V = L^-1 + K^-1 = V^T V) over the wall time of the step
        "roofline": {"kernel": "factor chain of the LML+gradient evaluations (potrf + V = L^-1 + K^-1 = V^T V), "
                               "all concurrent contexts of rank 0" + (" (average per GPU of the process)" if group else ""),
                     "bound": "mfma", "achieved": evals_gpu * float(Np) ** 3 / (elapsed / K) / 1e12,
                     "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": evals_gpu * float(Np) ** 3 / (elapsed / K) / 1e12 / F64_MFMA_PEAK_TFLOPS,
                     "traffic": None, "flops_per_call": float(Np) ** 3,
                     "evals_per_step_rank0": evals_rank, "ms_per_eval_wall": elapsed / K * 1e3 / max(evals_rank, 1),
                     "single_stage_chain_ms_under_contention": chain_ms,
                     "potrf_ms_under_contention": po_ms / max(po_n, 1)},
    }
    if comm is not None:
        comm.barrier()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_config1(args):
    """BASELINE configs[1]: 8-d correlated-Gaussian posterior, N_train = 1024, anisotropic RBF, one MI355X: kernel build +
    Cholesky (+ V = L^-1, alpha_) + posterior mean and std over M candidates (gpry/gpr.py:996-1020 `_update_model`,
    :1022-1273 `predict(return_std=True)`).  A step = one `_update_model()` at fixed theta (full rebuild: K, L, V, alpha_)
    followed by one pass of the predict kernels over the M candidates RESIDENT on the device (the sweep path: the same
    cross-kernel panel, contraction and finish kernels `predict` runs, results left in HBM); `predict()` through the host
    boundary -- candidates uploaded, mean / std copied back -- is timed beside it and is never `value`."""
    from gpry_amd.kernels import clone
    N, d, K, W = args.N or 1024, args.d or 8, args.steps, args.warmup
    M = args.M if args.M != 1_000_000 else 100_000
    bounds, X, y, Xc, truth = synthetic(N, d, M)
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
    gpr = GaussianProcessRegressor(kernel="RBF", bounds=bounds, noise_level=1e-2, preprocessing_X=Normalize_bounds(bounds),
                                   preprocessing_y=Normalize_y(), account_for_inf=None, verbose=1, random_state=3)
    theta = np.log(np.array([4.0] + [0.3] * d))            # fixed-theta stages, well conditioned
    k = clone(gpr.kernel)
    k.theta = theta
    gpr.kernel_, gpr._fitted = k, True
    gpr.append_to_data(X, y, fit_gpr=False)
    dev = gpr.device
    sigma_n, zeta = 1e-2, d ** -0.85
    mean0, std0 = gpr.predict(Xc, return_std=True)          # also sizes the device buffers
    dev.sweep_logexp(Xc, zeta, float(gpr.y_max), sigma_n, want=())      # the pool becomes resident

    def step():
        gpr.newly_appended_for_inv = 1
        gpr._update_model()
        gpr._push_affine()
        dev.sweep_logexp(None, zeta, float(gpr.y_max), sigma_n, M=M, want=())

    for _ in range(W):
        step()
    dev.sync()
    dev.timing_reset()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    dev.sync()
    elapsed = time.perf_counter() - t0
    names = ("kernel_build", "potrf", "trtri", "cross_build", "sweep_gemm", "sweep_gemm_splitk", "sweep_finish")
    T = {k_: dev.timing(k_) for k_ in names}
    dev.set_option("timing", 0)
    res = dev.sweep_fetch(want=("y", "sigma"))
    same = {"bit_identical": bool(np.array_equal(res["y"], mean0) and np.array_equal(res["sigma"], std0)),
            "max_abs_mean": float(np.max(np.abs(res["y"] - mean0))), "max_abs_std": float(np.max(np.abs(res["sigma"] - std0)))}
    # through the host boundary
    t1 = time.perf_counter()
    for _ in range(3):
        gpr.predict(Xc, return_std=True)
    predict_host_ms = (time.perf_counter() - t1) / 3 * 1e3
    Np = (N + 127) // 128 * 128
    per = lambda k_: T[k_][0] / max(T[k_][1], 1)            # ms per launch
    per_step = lambda k_: T[k_][0] / K
    kb_bytes = 8.0 * N * d + 8.0 * N * N
    kb_gbps = kb_bytes / (per("kernel_build") * 1e-3) / 1e9 if T["kernel_build"][1] else 0.0
    fac_flops = 2.0 * float(N) ** 3 / 3.0                   # potrf + V = L^-1
    fac_ms = per_step("potrf") + per_step("trtri")
    gemm_key = "sweep_gemm" if T["sweep_gemm"][1] else "sweep_gemm_splitk"
    launches = T[gemm_key][1]
    flops_launch = (float(N) * N + 2.0 * N) * M * K / max(launches, 1)
    achieved = flops_launch / (per(gemm_key) * 1e-3) / 1e12 if launches else 0.0
    cb_bytes = 8.0 * Np * M * K / max(T["cross_build"][1], 1)
    result = {
        "metric": "gp_update_model_plus_predict_throughput", "value": M / (elapsed / K), "unit": "candidates/s",
        "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {d}-d correlated-Gaussian posterior, N_train={N}, anisotropic RBF: kernel build "
                               f"+ Cholesky + V = L^-1 + predict(return_std=True) over M={M} resident candidates, fixed theta",
                   "N_train": N, "Np": Np, "d": d, "M_total": M, "kernel": "ConstantKernel*RBF", "rccl_ranks": 0},
        "stage_ms_per_step": {k_: per_step(k_) for k_ in names if T[k_][1]},
        "results_equal_predict": same, "predict_through_host_boundary_ms": predict_host_ms,
        "roofline": {"kernel": "sweep contraction (V lower-triangular x K*^T panel, sum-of-squares epilogue)", "bound": "mfma",
                     "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F64_MFMA_PEAK_TFLOPS,
                     "traffic": measured_traffic("config1", Np, M)[0], "traffic_note": measured_traffic("config1", Np, M)[1],
                     "avg_launch_ms": per(gemm_key), "launches": launches, "flops_per_launch": flops_launch},
        "kernel_build": {"bound": "hbm", "achieved": kb_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": kb_gbps / HBM_PEAK_GBPS,
                         "avg_launch_ms": per("kernel_build"), "launches": T["kernel_build"][1], "bytes_per_launch": kb_bytes},
        "factor": {"bound": "mfma", "achieved": fac_flops / (fac_ms * 1e-3) / 1e12 if fac_ms else 0.0, "peak": F64_MFMA_PEAK_TFLOPS,
                   "unit": "TFLOP/s", "frac": (fac_flops / (fac_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS) if fac_ms else 0.0,
                   "potrf_ms": per_step("potrf"), "trtri_ms": per_step("trtri"), "flops_per_call": fac_flops},
        "cross_build": {"bound": "hbm", "achieved": cb_bytes / (per("cross_build") * 1e-3) / 1e9 if T["cross_build"][1] else 0.0,
                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "avg_launch_ms": per("cross_build"), "bytes_per_launch": cb_bytes},
    }
    result["cross_build"]["frac"] = result["cross_build"]["achieved"] / HBM_PEAK_GBPS
    if args.cpu_baseline == "auto":
        try:
            from oracle import gpry_oracle as orc
            from threadpoolctl import threadpool_info
            pools = threadpool_info()
            threads = max([p_.get("num_threads", 1) for p_ in pools] or [1])
            m = orc.OracleGPR(bounds, kernel_id=orc.RBF)
            m.theta = theta
            m.fitted = True
            m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)       # warm-up
            t0 = time.perf_counter()
            m.newly_appended = 1
            m._update_model()
            t_upd = time.perf_counter() - t0
            n_s = min(M, CPU_CHUNK)
            t0 = time.perf_counter()
            rm, rs = m.predict(Xc[:n_s], return_std=True)
            t_pred = (time.perf_counter() - t0) / n_s
            C_ = np.exp(theta[0]) * m.pre_y.std_ ** 2
            result["parity_vs_port"] = {"mean_max_abs_rel": float(np.max(np.abs(mean0[:n_s] - rm)) / max(1.0, np.max(np.abs(rm)))),
                                        "var_max_abs_over_C": float(np.max(np.abs(std0[:n_s] ** 2 - rs ** 2)) / C_), "rows": n_s}
            cyc = t_upd + M * t_pred
            result["cpu_baseline"] = {"value": M / cyc, "unit": "candidates/s", "cores": threads, "kind": "port",
                                      "sample": f"oracle/gpry_oracle.py on host, {threads} BLAS threads: 1 _update_model at N={N} "
                                                f"({t_upd * 1e3:.1f} ms), predict+std on one chunk of {n_s} candidates ({t_pred * 1e6:.2f} us "
                                                f"each); step = update + {M} candidates = {cyc:.2f} s",
                                      "cpu_model": _cpu_model(), "update_model_s": t_upd, "predict_us_per_candidate": t_pred * 1e6}
        except Exception as e:
            result["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": None, "kind": "port", "sample": f"failed: {e!r}"}
    print(json.dumps(result))


def visible_gpus():
    """GPUs this process tree can use, counted in a short-lived child: the launcher itself never
    initialises HIP (a process that has must not be the parent of a rank's exec on this pool)."""
    code = ("import sys; sys.path.insert(0, %r); from gpry_amd import _lib; print('GPUS', _lib.device_count())" % ROOT)
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        for line in reversed(out.stdout.splitlines()):
            if line.startswith("GPUS "):
                return int(line.split()[1])
        print(f"bench.py: device probe failed: {out.stderr.strip()[-400:]}", file=sys.stderr)
    except Exception as e:
        print(f"bench.py: device probe failed: {e!r}", file=sys.stderr)
    return 0


def check_gpu_count(n):
    """Refuse ``--gpus n`` beyond the visible devices (round 2 ran 1 GPU and printed ``n_gpus: 1``)."""
    have = visible_gpus()
    if n <= have:
        return have
    if os.environ.get("GPRY_HIP_DEVICE_WRAP", "") == "1" and have >= 1:
        print(f"bench.py: --gpus {n} on {have} visible GPU(s): ranks share devices (GPRY_HIP_DEVICE_WRAP=1; "
              "no RCCL communicator can hold a device twice)", file=sys.stderr)
        return have
    print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; refusing (GPRY_HIP_DEVICE_WRAP=1 lets "
          "ranks share a device on development boxes)", file=sys.stderr)
    raise SystemExit(2)


def launch_ranks(n, argv, script=None, timeout=None, env_extra=None):
    """Parent of an N-GPU run: start one child process per GPU with the environment
    ``torch.distributed.run`` would give it, relay rank 0's stdout, return the worst exit code.

    The parent has not imported ``gpry_amd`` or touched HIP, and never ``exec``s: the ranks are ordinary
    children in their own sessions, so the watchdog can end a whole rank (its process group) when a peer
    died, when the time limit passed, or when the parent itself is told to stop."""
    import signal
    import socket
    import threading
    script = script or os.path.abspath(__file__)
    timeout = float(timeout if timeout is not None else os.environ.get("GPRY_BENCH_LAUNCH_TIMEOUT", "3600"))
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    import collections
    import tempfile
    progress = os.path.join(tempfile.mkdtemp(prefix="gpry_bench_"), "rank0_progress.json")
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "GPRY_BENCH_LAUNCHER": "self",
                    "GPRY_BENCH_PROGRESS_FILE": progress,
                    "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    # every rank's stderr is passed on as it comes and its tail kept: if a rank dies, the line says why
    tails = [collections.deque(maxlen=12) for _ in range(n)]

    def relay(r):
        for ln in procs[r].stderr:
            tails[r].append(ln.rstrip("\n"))
            sys.stderr.write(ln)
    relays = [threading.Thread(target=relay, args=(r,), daemon=True) for r in range(n)]
    for th in relays:
        th.start()

    def end_all(sig=signal.SIGTERM):
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, sig)        # the rank's own session: exactly the processes it started
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, frame):
        end_all(signal.SIGTERM)
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    t0, first_fail, why, t_end = time.time(), None, None, None
    fail_order = []          # ranks in the order in which they were seen to have failed on their own
    try:
        while any(pr.poll() is None for pr in procs):
            time.sleep(0.2)
            now = time.time()
            for r, pr in enumerate(procs):
                if t_end is None and pr.poll() not in (None, 0) and r not in fail_order:
                    fail_order.append(r)
            if first_fail is None and any(pr.poll() not in (None, 0) for pr in procs):
                first_fail = now
            if why is None and first_fail is not None and now - first_fail > float(os.environ.get("GPRY_BENCH_FAIL_GRACE", "30")):
                why = "a rank failed"            # its peers are waiting at a barrier that will never complete
            if why is None and now - t0 > timeout:
                why = f"time limit of {timeout:.0f} s"
            if why is not None and t_end is None:
                t_end = now
                end_all()
            if t_end is not None and now - t_end > 20.0:
                end_all(signal.SIGKILL)
    finally:
        end_all(signal.SIGKILL)
        for sg, h in old.items():
            signal.signal(sg, h)
    reader.join(timeout=10)
    for th in relays:
        th.join(timeout=5)
    codes = [pr.returncode for pr in procs]
    # the worst code a rank returned itself; ranks the launcher had to end (negative = signal) only count if no
    # rank failed on its own
    own = [c for c in codes if c is not None and c > 0]
    worst = max(own) if own else max([128 - c for c in codes if c is not None and c < 0] or [0])
    if fail_order and codes[fail_order[0]] is not None and codes[fail_order[0]] > 0:
        worst = codes[fail_order[0]]                          # the code of the rank that went down first
    if why is not None:
        print(f"bench.py: launcher ended the run ({why}); rank exit codes {codes}", file=sys.stderr)
        worst = worst or 124
    out = [ln.rstrip("\n") for ln in lines if ln.strip()]
    json_lines = [ln for ln in out if ln.lstrip().startswith("{")]
    for ln in out:
        if not ln.lstrip().startswith("{"):
            print(ln, file=sys.stderr)
    failed = [r for r, c in enumerate(codes) if c not in (0, None)]
    for r in failed:                                          # (a run that ended between two polls)
        if codes[r] is not None and codes[r] > 0 and r not in fail_order and t_end is None:
            fail_order.append(r)
    error = None
    if failed or why is not None:
        # the first rank that went down is the cause; whoever followed (a peer whose collective broke, the ranks the
        # launcher ended) is listed behind it
        first = fail_order[:1]
        error = {"failed_ranks": first or failed, "failed_in_order": fail_order, "exit_codes": codes, "launcher": why,
                 "stderr_tail": {str(r): list(tails[r])[-6:] for r in (first or failed)[:4]}}
    line = None
    if json_lines:
        line = json_lines[-1]
        if error is not None:           # rank 0 reported, a peer failed afterwards: the line keeps its numbers and says so
            try:
                obj = json.loads(line)
                obj["error"] = error
                line = json.dumps(obj)
            except ValueError:
                pass
    elif os.path.exists(progress):      # rank 0 never got to its line: its last snapshot, marked as such
        try:
            obj = json.load(open(progress))
            obj["error"] = error or {"failed_ranks": [], "exit_codes": codes, "launcher": "rank 0 printed no result line"}
            line = json.dumps(obj)
        except (OSError, ValueError):
            line = None
    if line is not None:
        print(line, flush=True)
    elif worst == 0:
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        worst = 1
    try:
        os.remove(progress)
        os.rmdir(os.path.dirname(progress))
    except OSError:
        pass
    return worst


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cycle", "farm", "config1"], default="cycle",
                    help="cycle: BASELINE configs[2] (the headline; configs[3] with --gpus N); farm: configs[4]; config1: configs[1]")
    ap.add_argument("--mode", choices=["ranks", "group"], default="ranks",
                    help="farm: 'ranks' = one process per GPU (gpry/run.py:1238-1293 over RCCL), 'group' = ONE process "
                         "whose fit contexts are spread over --gpus devices (a single-process Runner)")
    ap.add_argument("--N", type=int, default=None)
    ap.add_argument("--d", type=int, default=None)
    ap.add_argument("--M", type=int, default=1_000_000)
    ap.add_argument("--n-points", type=int, default=None)
    ap.add_argument("--restarts", type=int, default=32)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong")
    ap.add_argument("--allow-gloo", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["auto", "off"], default="auto")
    ap.add_argument("--extras", choices=["auto", "off"], default="auto")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    group_mode = args.workload == "farm" and args.mode == "group"
    if args.mode == "group" and args.workload != "farm":
        raise SystemExit("bench.py: --mode group applies to --workload farm (the cycle shards in one process through "
                         "NORA(devices=...), tools/bench_group.py)")
    if group_mode and launched and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        raise SystemExit("bench.py: --mode group is ONE process; do not start it under torch.distributed.run")
    if not launched and args.gpus > 1:
        have = check_gpu_count(args.gpus)              # exits 2 beyond the visible devices
        if group_mode:
            os.environ["GPRY_BENCH_GROUP_DEVICES"] = ",".join(str(i % max(have, 1)) for i in range(args.gpus))
        else:
            # this process becomes the launcher: it has not imported the library or touched HIP
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not group_mode:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (self-"
              f"launching) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`", file=sys.stderr)
        raise SystemExit(2)
    os.environ["GPRY_HIP_DEVICE"] = str(local_rank)
    if args.workload == "farm":
        return run_farm(args, rank, world, local_rank)
    if args.workload == "config1":
        if world != 1:
            raise SystemExit("bench.py: --workload config1 is a one-GPU configuration")
        return run_config1(args)

    from gpry_amd.gp_acquisition import NORA

    N, d, K, W = args.N or 4096, args.d or 16, args.steps, args.warmup
    npts = args.n_points or d
    M_total = args.M * world if args.scaling == "weak" else args.M
    n_base = N - npts
    if n_base < 2 * d:
        raise SystemExit("N too small")
    bounds, X, y, Xc, truth = synthetic(n_base, d, M_total)
    # "multi_add with a FRESH candidate pool" (gpry/gp_acquisition.py:1023-1031, 858-873: every mc_every-th
    # call draws a new X_mc).  Two pre-generated pools handed out in turn: a step never sees the array object of the step
    # before, so the pool is uploaded (128 MB at M = 1e6, d = 16) inside every timed step -- chunk by chunk underneath the
    # sweep (option "sweep_upload").  The same loop on ONE pool, which stays resident in HBM, is timed behind it and
    # reported as cycle.resident_pool_ms_per_step.
    pools = [Xc, synthetic(n_base, d, M_total, seed_cand=101)[3]]
    pool_turn = [0]

    def fresh_pool(gpr, bounds=None, rng=None, sampler=None):
        pool_turn[0] += 1
        return pools[pool_turn[0] % 2], None, None, None

    gpr = make_gpr(bounds)
    dev = gpr.device
    comm, dist, comm_kind, n_rccl = None, None, "none", 0
    if world > 1 or "RANK" in os.environ:   # launched by torch.distributed.run (also with 1 rank)
        comm, dist, comm_kind, n_rccl = connect(args, rank, world, dev)

    if rank == 0 and world > 1:
        M_each = [int(min(M_total, (r + 1) * -(-M_total // world)) - min(M_total, r * -(-M_total // world))) for r in range(world)]
        progress_snapshot({"metric": "gp_refit_plus_nora_acq_cycle_throughput", "value": None, "unit": "candidates/s", "n_gpus": world,
                           "config": {"workload": "BASELINE configs[3]", "N_train": N, "d": d, "M_total": M_total,
                                      "M_per_gpu_by_rank": M_each, "comm": comm_kind, "rccl_ranks": n_rccl}}, "after the rendezvous")
    # --gpus N means N processes with one GPU each: a 1-process run must not pick up the other GPUs of
    # the node through NORA's in-process device group (its default is every visible GPU)
    sharded = comm is not None and world > 1
    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, comm=comm,
               devices=None if sharded else [local_rank])
    acq.do_MC_sample = fresh_pool
    rng = np.random.default_rng(2)

    # ---- setup (untimed): first fit of the base set, first proposals, pool upload
    t0 = time.perf_counter()
    e0 = gpr.n_eval_loglike
    gpr.append_to_data(X, y, fit_gpr="simple")
    first_fit = {"ms": (time.perf_counter() - t0) * 1e3, "lml_grad_evals": gpr.n_eval_loglike - e0,
                 "N_train": gpr.n, "note": "unfitted model: one run from a random start (gpry/gpr.py:917-918), "
                                           "includes context creation and allocation"}
    X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)

    host_t = {"refit": 0.0, "acq": 0.0}
    n_seen = []
    work = {"sweep_flops": 0.0, "kb_bytes": 0.0, "potrf_flops_each": []}

    def step(acq=acq):
        nonlocal X_new
        t_a = time.perf_counter()
        rewind(gpr, n_base)
        gpr.append_to_data(X_new, truth(X_new), fit_gpr="simple")
        t_b = time.perf_counter()
        X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)
        host_t["refit"] += t_b - t_a
        host_t["acq"] += time.perf_counter() - t_b
        n = gpr.n
        n_seen.append(n)
        # algorithmic work of THIS step at the size it actually ran at
        work["sweep_flops"] += (acq._sweep_hi - acq._sweep_lo) * (float(n) ** 2 + 2.0 * n)

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
    n_seen.clear()
    work["sweep_flops"] = 0.0
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
    names = ("kernel_build", "potrf", "trtri", "lauum", "lml_traces", "cross_build", "sweep_gemm",
             "sweep_finish", "topk")
    T = {k: dev.timing(k) for k in names}
    sweep_flops_timed, n_seen_timed, host_timed = work["sweep_flops"], list(n_seen), dict(host_t)
    # ---- the same cycle on ONE pool that stays resident in HBM (rounds 1-4 timed this): min(K, 5) steps behind the
    # headline loop, every rank alike
    # (one-GPU runs only: behind the timed region of an N-GPU run nothing collective may follow before rank 0 has printed
    # its line -- a peer that dies there must not take the measurement with it)
    K_res, resident_elapsed = 0, None
    if world == 1:
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (pools[0], None, None, None)
        step()                                          # (the first one uploads pools[0])
        fence()
        K_res = min(K, 5)
        t0 = time.perf_counter()
        for _ in range(K_res):
            step()
        fence()
        resident_elapsed = time.perf_counter() - t0
        acq.do_MC_sample = fresh_pool
    work["sweep_flops"] = sweep_flops_timed
    n_seen[:] = n_seen_timed
    host_t.update(host_timed)
    if any(n != N for n in n_seen):
        raise SystemExit(f"bench.py: a timed step did not run at N_train={N}: {n_seen}")

    per_step_ms = {k: T[k][0] / K for k in names}
    Np = (N + 127) // 128 * 128
    M_rank = acq._sweep_hi - acq._sweep_lo
    # dominant kernel: the FP64-MFMA triangular GEMM of the sweep.  Algorithmic flops per candidate =
    # N^2 (mul+add over the N^2/2 non-zeros of V) + 2N (mean), SURVEY.md 8(d), summed over the steps at
    # the N each one ran at, over the HIP-event time of exactly those launches
    gemm_ms, gemm_n = T["sweep_gemm"]
    flops_launch = work["sweep_flops"] / max(gemm_n, 1)
    achieved = work["sweep_flops"] / (gemm_ms * 1e-3) / 1e12 if gemm_ms else 0.0
    traffic, traffic_note = measured_traffic("config2", Np, M_rank)
    kb_ms, kb_n = T["kernel_build"]
    kb_bytes = 8.0 * N * N + 8.0 * N * d          # every launch ran at N (checked above)
    kb_gbps = kb_bytes / (kb_ms / max(kb_n, 1) * 1e-3) / 1e9 if kb_ms else 0.0
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
        "config": {"workload": ("BASELINE configs[2]" if world == 1 else "BASELINE configs[3]") +
                               ": 16-d correlated-Gaussian posterior, N_train=4096 in every step, "
                               "Matern-5/2, LogExp NORA sweep M=1e6, n_points=16, fit_gpr='simple'" +
                               ("" if world == 1 else f", candidate pool sharded {world}-way, shortlist all-gather over "
                                + ("RCCL" if comm_kind == "rccl" else "gloo (RCCL unavailable, --allow-gloo)")),
                   "N_train": N, "N_train_per_step": list(n_seen), "Np": Np, "d": d, "M_total": M_total,
                   "M_per_gpu": M_rank, "n_points": npts,
                   "kernel": "ConstantKernel*Matern(nu=2.5)", "sharding": f"candidates x{world}",
                   "comm": comm_kind, "rccl_ranks": n_rccl},
        "cycle": {"refit_plus_acq_ms": ms_per_step,
                  # the headline (value, ms_per_step) IS the fresh-pool cycle: every timed step uploads its 8 M d bytes of
                  # candidates; the resident-pool figure is what rounds 1-4 reported
                  "fresh_pool_ms_per_step": ms_per_step,
                  "resident_pool_ms_per_step": resident_elapsed / K_res * 1e3 if K_res else None,
                  "resident_pool_steps": K_res,
                  "pool_upload": {"bytes_per_step": 8.0 * M_rank * d, "where": "chunk by chunk on a copy stream, chunk c + 1 "
                                  "underneath the kernels of chunk c (library option sweep_upload)",
                                  "pipelined": bool(dev.get_option("sweep_upload")) if hasattr(dev, "get_option") else None},
                  "refit_ms": host_t["refit"] / K * 1e3,
                  "acquisition_ms": host_t["acq"] / K * 1e3,
                  "one_lml_grad_call_ms": (per_step_ms["kernel_build"] + per_step_ms["potrf"] + per_step_ms["trtri"] +
                                           per_step_ms["lauum"] + per_step_ms["lml_traces"]) / max(lml_evals, 1.0),
                  "lml_grad_evals_per_step": lml_evals,
                  "stage_ms_per_step": per_step_ms, "device_sweep_ms_per_step": sweep_ms,
                  "sweep_candidates_per_s_per_gpu": M_rank / (sweep_ms * 1e-3) if sweep_ms else None,
                  # which form of the cross-kernel panel the fitted model's error estimates allowed (gpry_sweep_info)
                  "panel_form": acq.stats.get("panel_form"), "panel_error_estimate": acq.stats.get("panel_error_estimate"),
                  "panel_error_mean_worst_case": acq.stats.get("panel_error_mean_worst_case"),
                  "panel_error_variance": acq.stats.get("panel_error_variance"), "panel_gate": acq.stats.get("panel_gate"),
                  "shortlist": acq.stats.get("shortlist"), "cache_models_per_step": cache_models / K,
                  "rank_host_ms": acq.stats.get("rank_s", 0) * 1e3,
                  "first_fit_untimed": first_fit},
        "roofline": {"kernel": "sweep_gemm_dma_sp_kernel (V lower-triangular x K*^T panel, sum-of-squares epilogue)",
                     "bound": "mfma", "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "traffic_note": traffic_note,
                     "avg_launch_ms": gemm_ms / max(gemm_n, 1), "launches": gemm_n,
                     "flops_per_launch": flops_launch},
        "kernel_build": {"bound": "hbm", "achieved": kb_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": kb_gbps / HBM_PEAK_GBPS, "avg_launch_ms": kb_ms / max(kb_n, 1),
                         "launches": kb_n, "bytes_per_launch": kb_bytes},
    }
    # MFMA utilisation of the Cholesky (north_star): Np^3/3 flop per factorisation over the average
    # duration of the `potrf` stage, and of the whole factor chain of one LML+gradient call
    # (potrf + V = L^-1 + K^-1 = V^T V: Np^3 flop); every call ran at N (checked above)
    po_ms, po_n = T["potrf"]
    if po_n:
        tf = (float(Np) ** 3 / 3.0) / (po_ms / po_n * 1e-3) / 1e12
        chain_ms = sum(T[k][0] / max(T[k][1], 1) for k in ("potrf", "trtri", "lauum"))
        result["cholesky"] = {"bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": tf / F64_MFMA_PEAK_TFLOPS, "avg_ms": po_ms / po_n, "calls": po_n,
                              "flops_per_call": float(Np) ** 3 / 3.0,
                              "factor_chain_tflops": float(Np) ** 3 / (chain_ms * 1e-3) / 1e12}
    if rank == 0 and po_n:
        # From Np = 4096 on the first three quarters of V = L^-1 run on a second stream underneath the
        # panel chain of potrf (csrc/chol.hip: trtri_pipeline_*): the `potrf` stage above then contains
        # the price of sharing the GPU, the `trtri` stage only what is left after it.  Quote the same
        # evaluation with the serial chain beside it (outside the timed region).
        try:
            import time as _time
            th = np.array(gpr.kernel_.theta, dtype=float)
            serial = {}
            for pipe in (0, 1):
                dev.set_option("factor_pipeline", pipe)
                for _ in range(2):
                    dev.lml(th, True)
                dev.timing_reset()
                t0 = _time.perf_counter()
                for _ in range(8):
                    dev.lml(th, True)
                wall = (_time.perf_counter() - t0) / 8 * 1e3
                serial[pipe] = {"lml_grad_wall_ms": wall,
                                **{k + "_ms": dev.timing(k)[0] / max(dev.timing(k)[1], 1) for k in ("potrf", "trtri", "lauum")}}
            dev.set_option("timing", 0)
            result["cholesky"]["serial_chain"] = serial[0]
            result["cholesky"]["pipelined_chain"] = serial[1]
            result["cholesky"]["potrf_alone_tflops"] = (float(Np) ** 3 / 3.0) / (serial[0]["potrf_ms"] * 1e-3) / 1e12
        except Exception as e:
            result["cholesky"]["serial_chain_error"] = repr(e)
        finally:
            dev.set_option("factor_pipeline", 1)
    if rank == 0:
        # the build is one 40-us launch per LML evaluation: two events around a single launch also
        # time the dispatch gap, so quote the steady-state duration (50 launches back to back) too
        try:
            us = dev.microbench(6, 50)
            result["kernel_build"].update({
                "avg_launch_ms_back_to_back": us * 1e-3,
                "achieved_back_to_back": kb_bytes / (us * 1e-6) / 1e9,
                "frac_back_to_back": kb_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS})
        except Exception as e:
            result["kernel_build"]["back_to_back_error"] = repr(e)
        # measured ceilings of this very GPU, quoted beside the spec peaks used for `frac`
        try:
            mp = {"mfma_f64_vgpr_acc_TFLOPs": dev.microbench(2, 1),
                  "hbm_copy_GBps": dev.microbench(1, 1 << 30),
                  "hbm_fill_GBps": dev.microbench(3, 1 << 30)}
            result["measured_peaks"] = mp
            result["roofline"]["frac_of_measured_peak"] = achieved / mp["mfma_f64_vgpr_acc_TFLOPs"]
            if achieved > 1.03 * mp["mfma_f64_vgpr_acc_TFLOPs"]:
                result["roofline"]["warning"] = ("achieved exceeds the MFMA rate measured on this GPU by "
                                                 "more than 3 %: check the flop accounting")
        except Exception as e:
            result["measured_peaks"] = {"error": repr(e)}

    if rank == 0 and world > 1:
        progress_snapshot(result, "after the timed region (scaling_check pending)")
    # ---- strong scaling: the same cycle on ONE GPU of this node, measured in this very run, so
    # that the line carries its own speed-up (whole cycle and device sweep)
    if world > 1 and args.scaling == "strong" and args.extras == "auto" and rank == 0:
        try:      # an exception here must not leave the other ranks waiting at the barrier below
            acq1 = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, comm=None, devices=[local_rank])
            acq1.do_MC_sample = acq.do_MC_sample
            X_new, _, _ = acq1.multi_add(gpr, n_points=npts, rng=rng)    # uploads the whole pool
            step(acq1)
            dev.sync()
            dev.timing_reset()
            t1 = time.perf_counter()
            for _ in range(2):
                step(acq1)
            dev.sync()
            one_ms = (time.perf_counter() - t1) / 2 * 1e3
            one_sweep = sum(dev.timing(k)[0] for k in ("cross_build", "sweep_gemm", "sweep_finish")) / 2
            result["scaling_check"] = {
                "one_gpu_ms_per_step": one_ms, "one_gpu_device_sweep_ms": one_sweep,
                "cycle_speedup": one_ms / ms_per_step,
                "device_sweep_speedup": one_sweep / sweep_ms if sweep_ms else None,
                "note": "1-GPU figures: rank 0 alone, whole pool, 2 steps after the timed region"}
        except Exception as e:
            result["scaling_check"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.extras == "auto":
        try:
            result["refit_extras"] = refit_extras(bounds, gpr.X_train_all.copy(), gpr.y_train_all.copy())
        except Exception as e:
            result["refit_extras"] = {"error": repr(e)}
        # The reference's own refit cadence (gpry/run.py:317, :531, :1250-1256): a full fit of 10 + 2 d restarts every
        # fit_full_every = 2 sqrt(d) iterations, the one-run "simple" refit in between.  The full-fit iteration = the timed
        # cycle with its simple refit replaced by the full fit measured above (warm contexts); the headline stays the
        # simple cycle (BASELINE.json's metric).
        try:
            ff = result["refit_extras"]["full_fit_default_restarts"]
            every = max(1, int(round(2.0 * math.sqrt(d))))
            full_ms = ff.get("ms_warm", ff["ms"])
            full_cycle = ms_per_step - result["cycle"]["refit_ms"] + full_ms
            result["cycle"]["runner_cadence"] = {
                "fit_full_every": every, "n_restarts": ff["n_restarts"], "full_fit_ms": full_ms,
                "full_fit_lml_grad_evals": ff.get("lml_grad_evals_warm", ff["lml_grad_evals"]),
                "full_fit_iteration_ms": full_cycle, "simple_iteration_ms": ms_per_step,
                "note": "((fit_full_every - 1) x simple cycle + 1 x (cycle with the full fit in place of the simple refit)) / fit_full_every"}
            result["cycle"]["runner_cadence_ms"] = ((every - 1) * ms_per_step + full_cycle) / every
        except Exception as e:
            result["cycle"]["runner_cadence"] = {"error": repr(e)}
        try:
            result["small_n"] = small_n_extras()
        except Exception as e:
            result["small_n"] = {"error": repr(e)}
    if rank == 0 and world == 1 and args.cpu_baseline == "auto":
        try:
            result["cpu_baseline"] = cpu_baseline(N, d, args.M, lml_evals, cache_models / K)
        except Exception as e:   # the baseline must never take the GPU number down with it
            result["cpu_baseline"] = {"value": None, "unit": "candidates/s", "cores": None,
                                      "kind": "port", "sample": f"failed: {e!r}"}
    if rank == 0 and isinstance(result.get("small_n"), dict) and isinstance(result.get("cpu_baseline", {}).get("small_n"), dict):
        for key, row in result["small_n"].items():       # full fit on the host where it was not run: evaluations x time per evaluation
            cpu_row = result["cpu_baseline"]["small_n"].get(key)
            if isinstance(row, dict) and "fit_full_evals" in row and cpu_row and "fit_full_s" not in cpu_row:
                cpu_row["fit_full_s_priced"] = row["fit_full_evals"] * cpu_row["lml_grad_us"] * 1e-6
    # rank 0 prints BEFORE the closing barriers: a rank that dies late (after the timed region) then cannot take the
    # line with it -- rank 0 needs no further collective to report
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()      # gloo: the other ranks wait here while rank 0 measures its extras
        dist.barrier()
        dist.destroy_process_group()


def progress_snapshot(result, stage):
    """Self-launched N-GPU runs: rank 0 leaves what it knows so far (configuration incl. ``rccl_ranks`` and the shard sizes
    after the rendezvous; the measured cycle after the timed region) in the file the launcher named.  If a rank fails
    before rank 0 prints its line, the launcher prints the last snapshot with an ``error`` object that says which rank
    and why (``launch_ranks``)."""
    path = os.environ.get("GPRY_BENCH_PROGRESS_FILE", "")
    if not path:
        return
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(dict(result, partial=stage), f)
        os.replace(path + ".tmp", path)
    except OSError:
        pass


if __name__ == "__main__":
    main()
