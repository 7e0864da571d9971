#!/usr/bin/env python3
"""Verbose GPU self-test of libgpry_hip.so against the CPU oracle (run via gpurun).

Prints per-stage maximum errors and stage timings so that one GPU call gives enough
information to locate a wrong kernel.  Not part of the pytest suite (tests/ has the
pass/fail versions); writes gpurun_out/selftest.json.
"""
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpry_amd import _lib  # noqa: E402
from oracle import gpry_oracle as orc  # noqa: E402

RESULTS = {}


def report(name, **kw):
    RESULTS[name] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()}
    print(f"[{name}] " + " ".join(f"{k}={v:.3e}" if isinstance(v, (float, np.floating)) else f"{k}={v}"
                                  for k, v in kw.items()), flush=True)


def relmax(a, b):
    a, b = np.asarray(a), np.asarray(b)
    den = np.max(np.abs(b)) + 1e-300
    return float(np.max(np.abs(a - b)) / den)


def test_gemm(dev):
    rng = np.random.default_rng(0)
    for (M, N, K) in ((128, 128, 64), (256, 192, 128), (64, 64, 64)):
        A = rng.standard_normal((M, K))
        B = rng.standard_normal((K, N))
        ref = A @ B
        for at in (0, 1):
            for bt in (0, 1):
                Ain = np.ascontiguousarray(A.T) if at else A
                Bin = np.ascontiguousarray(B.T) if bt else B
                C = dev.debug_gemm(Ain, Bin, None, M, N, K, at, bt, epi=0)
                report(f"gemm_store_{M}x{N}x{K}_at{at}_bt{bt}", err=relmax(C, ref))
        C0 = rng.standard_normal((M, N))
        C = dev.debug_gemm(A, B, C0, M, N, K, epi=2)
        report(f"gemm_sub_{M}x{N}x{K}", err=relmax(C, C0 - ref))
        C = dev.debug_gemm(A, B, None, M, N, K, epi=1)
        report(f"gemm_neg_{M}x{N}x{K}", err=relmax(C, -ref))
        C = dev.debug_gemm(A, B, None, M, N, K, epi=3)
        tiles = (M + 127) // 128
        ss = np.stack([np.sum(ref[t * 128:(t + 1) * 128] ** 2, axis=0) for t in range(tiles)])
        report(f"gemm_sumsq_{M}x{N}x{K}", err=relmax(C, ss))
    # triangular k-range modes on square problems
    n = 384
    L = np.tril(rng.standard_normal((n, n)))
    B = rng.standard_normal((n, n))
    C = dev.debug_gemm(L, B, None, n, n, n, kmode=1)          # A lower
    report("gemm_kmode_A_lower", err=relmax(C, L @ B))
    C = dev.debug_gemm(B, L, None, n, n, n, kmode=2)          # B lower
    report("gemm_kmode_B_lower", err=relmax(C, B @ L))
    C = dev.debug_gemm(L, L, None, n, n, n, a_trans=1, kmode=3, lower_only=1)  # L^T L, lower tiles
    ref = L.T @ L
    mask = np.tril(np.ones((n // 128, n // 128))).repeat(128, 0).repeat(128, 1).astype(bool)
    report("gemm_kmode_lauum", err=relmax(C[mask], ref[mask]))
    U = np.ascontiguousarray(L.T)
    C = dev.debug_gemm(B, L, None, n, n, n, b_trans=1, kmode=4)  # B(k,j)=L[j][k] upper
    report("gemm_kmode_B_upper", err=relmax(C, B @ U))
    C = dev.debug_gemm(L, B, None, n, 1024, n, kmode=1, epi=3, tile_map=1) if False else None
    B2 = rng.standard_normal((n, 1280))
    C = dev.debug_gemm(L, B2, None, n, 1280, n, kmode=1, epi=3, tile_map=1)
    ref = L @ B2
    ss = np.stack([np.sum(ref[t * 128:(t + 1) * 128] ** 2, axis=0) for t in range(n // 128)])
    report("gemm_sweepmap_sumsq", err=relmax(C, ss))


def model(N, d, kid, seed=0, M=256):
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed)
    g = orc.OracleGPR(bounds, kernel_id=kid)
    g.theta = np.log(np.array([4.0] + [0.3 + 0.02 * k for k in range(d)]))
    g.fitted = True
    g.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    return g, Xc


def setup_dev(dev, g):
    dev.set_train(g.X_train_, g.y_train_, g.alpha)
    dev.set_theta(g.kernel_id, g.theta)
    dev.set_affine(g.pre_X.lo, g.pre_X.hi - g.pre_X.lo, g.pre_y.mean_, g.pre_y.std_, g.clip_hi())


def test_model(dev, N, d, kid, do_lml=True, M=300):
    tag = f"N{N}_d{d}_k{kid}"
    g, Xc = model(N, d, kid, seed=N + d + kid, M=M)
    setup_dev(dev, g)
    K = dev.kernel_train(add_alpha=False)
    Kref = orc.kernel_matrix(g.X_train_, g.theta, kid)
    report(f"kernel_train_{tag}", err=relmax(K, Kref), sym=float(np.max(np.abs(K - K.T))))
    Xc_ = g.pre_X.transform(Xc)
    Kx = dev.kernel_cross(Xc_)
    report(f"kernel_cross_{tag}", err=relmax(Kx, orc.kernel_matrix(Xc_, g.theta, kid, Y=g.X_train_)))
    info = dev.factorize()
    L, V, a = dev.get_factor()
    report(f"factor_{tag}", info=info, L=relmax(L, g.L_), V=relmax(V, g.V_), alpha=relmax(a, g.alpha_),
           LLt=relmax(L @ L.T, Kref + np.diag(g.alpha)), VL=relmax(V @ L, np.eye(N)))
    if do_lml:
        th = g.theta + 0.1
        lml, grad, info = dev.lml(th, True)
        rl, rg = orc.log_marginal_likelihood(g.X_train_, g.y_train_, g.alpha, th, kid, True)
        report(f"lml_{tag}", info=info, lml=abs(lml - rl) / abs(rl), grad=relmax(grad, rg))
        lml2, _ = dev.lml(th, False)
        report(f"lml_nograd_{tag}", lml=abs(lml2 - rl) / abs(rl))
    mean, std = dev.predict(Xc, return_std=True)
    rm, rs = g.predict(Xc, return_std=True)
    C = np.exp(g.theta[0]) * g.pre_y.std_ ** 2
    report(f"predict_{tag}", mean=relmax(mean, rm), var_over_C=float(np.max(np.abs(std ** 2 - rs ** 2)) / C),
           std=relmax(std, rs))
    mean_only = dev.predict(Xc, return_std=False)
    report(f"predict_meanonly_{tag}", mean=relmax(mean_only, rm))
    zeta = orc.auto_zeta(d)
    out = dev.sweep_logexp(Xc, zeta, g.y_max, g.noise_level)
    racq = orc.logexp_f(rm, rs, g.y_max, g.noise_level, zeta)
    fin = np.isfinite(racq)
    report(f"sweep_{tag}", n_nan=out["n_nan"], acq=relmax(out["acq"][fin], racq[fin]),
           inf_match=bool(np.array_equal(np.isneginf(out["acq"]), np.isneginf(racq))),
           argmax_same=bool(np.argmax(out["acq"]) == np.argmax(racq)))
    top, bound = dev.sweep_topk(32)
    order = np.lexsort((-np.arange(len(out["acq"])), -out["acq"]))  # acq desc, idx desc
    report(f"topk_{tag}", same=bool(np.array_equal(top["idx"], order[:32])),
           bound_ok=bool(bound == out["acq"][order[32]]))
    top2, _ = dev.sweep_topk(16, exclude=order[:5])
    report(f"topk_excl_{tag}", same=bool(np.array_equal(top2["idx"], order[5:21])))
    # Kriging believer: u vectors and gram
    dev.kb_reset()
    first, var0 = dev.kb_register(Xc[:40])
    Kst = orc.kernel_matrix(Xc_[:40], g.theta, kid, Y=g.X_train_)
    U = (g.V_ @ Kst.T).T
    report(f"kb_var0_{tag}", err=float(np.max(np.abs(var0 - (np.exp(g.theta[0]) - np.sum(U ** 2, axis=1)))) / np.exp(g.theta[0])))
    G, kv = dev.kb_gram(7, 40)
    report(f"kb_gram_{tag}", G=relmax(G, U @ U[7]), k=relmax(kv, orc.kernel_matrix(Xc_[:40], g.theta, kid, Y=Xc_[7:8])[:, 0]))


def timing_run(dev, N=4096, d=16, kid=3, M=131072):
    g_bounds, X, y, Xc = orc.synthetic_problem(N, d, M)
    pre = orc.NormalizeBounds(g_bounds)
    X_ = pre.transform(X)
    ymean, ystd = y.mean(), y.std()
    y_ = (y - ymean) / ystd
    alpha = np.full(N, (1e-2 / ystd) ** 2)
    theta = np.log(np.array([4.0] + [0.3] * d))
    dev.set_train(X_, y_, alpha)
    dev.set_theta(kid, theta)
    dev.set_affine(pre.lo, pre.hi - pre.lo, ymean, ystd, 1.1 * y.max() - 0.1 * y.min())
    for rep in range(2):
        dev.timing_reset()
        t0 = time.time(); info = dev.factorize(); t_fac = time.time() - t0
        t0 = time.time(); lml, grad, info2 = dev.lml(theta, True); t_lml = time.time() - t0
        t0 = time.time(); out = dev.sweep_logexp(Xc, orc.auto_zeta(d), y.max(), 1e-2); t_sw = time.time() - t0
        t0 = time.time(); top, bound = dev.sweep_topk(1024); t_tk = time.time() - t0
        stages = {k: dev.timing(k) for k in ("kernel_build", "potrf", "trtri", "lauum", "lml_traces",
                                              "cross_build", "sweep_gemm", "sweep_finish", "topk")}
        report(f"timing_N{N}_rep{rep}", info=info, factorize_s=t_fac, lml_s=t_lml, sweep_s=t_sw, topk_s=t_tk,
               cand_per_s=M / t_sw, lml=lml,
               **{f"{k}_ms": v[0] for k, v in stages.items()}, **{f"{k}_n": v[1] for k, v in stages.items()})
    flops = M * (float(N) ** 2)
    report("sweep_gemm_rate", tflops=flops / (stages["sweep_gemm"][0] * 1e-3) / 1e12)
    report("kernel_build_rate", GBps=(8.0 * N * N + 8.0 * N * d) * stages["kernel_build"][1] /
           (stages["kernel_build"][0] * 1e-3) / 1e9)


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    dev = _lib.Device(0)
    print(dev.info(), flush=True)
    steps = [("gemm", lambda: test_gemm(dev))]
    for (N, d, kid, do_lml) in ((40, 3, 0, True), (200, 5, 3, True), (300, 2, 1, True), (333, 8, 2, True),
                                (1000, 16, 3, True), (1024, 8, 0, True)):
        steps.append((f"model_{N}_{d}_{kid}", lambda N=N, d=d, kid=kid, do_lml=do_lml: test_model(dev, N, d, kid, do_lml)))
    steps.append(("peaks", lambda: report("peaks", mfma_f64_tflops=dev.microbench(0), hbm_copy_GBps=dev.microbench(1, 1 << 30))))
    steps.append(("timing", lambda: timing_run(dev)))
    for name, fn in steps:
        try:
            fn()
        except Exception as e:  # keep going: one call should tell us as much as possible
            traceback.print_exc()
            RESULTS[f"EXC_{name}"] = repr(e)
    with open(os.path.join(ROOT, "gpurun_out", "selftest.json"), "w") as f:
        json.dump(RESULTS, f, indent=1, default=str)
    bad = [k for k in RESULTS if k.startswith("EXC_")]
    print("exceptions:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
