"""The drop-in boundary is a plain C ABI: a C program (gcc, no Python / C++ / torch) links
``libgpry_hip.so`` through ``include/gpry_hip.h`` and reproduces the oracle's numbers."""
import os
import subprocess

import numpy as np
import pytest

from oracle import gpry_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_client_of_the_shared_library(tmp_path):
    exe = tmp_path / "c_abi_smoke"
    lib_dir = os.path.join(ROOT, "gpry_amd")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.c"), "-o", str(exe),
                    "-L", lib_dir, "-l:libgpry_hip.so", f"-Wl,-rpath,{lib_dir}"], check=True)
    N, d, M, kid = 150, 3, 40, 3
    rng = np.random.default_rng(4)
    X = rng.uniform(0, 1, (N, d))
    y = np.sin(3 * X).sum(1)
    alpha = np.full(N, 1e-6)
    theta = np.log(np.array([2.0, 0.4, 0.5, 0.6]))
    Xc = rng.uniform(0, 1, (M, d))
    lines = [f"{N} {d} {M} {kid}", " ".join(repr(float(v)) for v in theta)]
    lines += [" ".join(repr(float(v)) for v in list(X[i]) + [y[i], alpha[i]]) for i in range(N)]
    lines += [" ".join(repr(float(v)) for v in Xc[m]) for m in range(M)]
    out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = [ln.split() for ln in out.stdout.strip().splitlines()]
    lml = float([r for r in rows if r[0] == "lml"][0][1])
    grad = np.array([float(v) for v in [r for r in rows if r[0] == "grad"][0][1:]])
    pred = np.array([[float(v) for v in r[1:]] for r in rows if r[0] == "pred"])
    top = [(int(r[1]), float(r[2])) for r in rows if r[0] == "top"]
    rl, rg = orc.log_marginal_likelihood(X, y, alpha, theta, kid, eval_gradient=True)
    assert abs(lml - rl) <= 1e-10 * abs(rl)
    np.testing.assert_allclose(grad, rg, rtol=1e-7, atol=1e-7 * np.max(np.abs(rg)))
    K = orc.kernel_matrix(X, theta, kid)
    K[np.diag_indices_from(K)] += alpha
    L, V, a = orc.factorize(K, y)
    Kt = orc.kernel_matrix(Xc, theta, kid, Y=X)
    mean = Kt @ a
    var = np.exp(theta[0]) - np.einsum("ij,ij->j", V @ Kt.T, V @ Kt.T)
    np.testing.assert_allclose(pred[:, 0], mean, rtol=1e-8, atol=1e-9)
    assert np.max(np.abs(pred[:, 1] ** 2 - np.clip(var, 0, None))) <= 1e-9 * np.exp(theta[0])
    acq = orc.logexp_f(mean, np.sqrt(np.clip(var, 0, None)), 0.0, 1e-3, 0.25)
    np.testing.assert_allclose(pred[:, 2], acq, rtol=1e-6, atol=1e-6)
    order = np.lexsort((-np.arange(M), -pred[:, 2]))
    assert [t[0] for t in top] == list(order[:5])


def test_error_returns_of_the_c_abi():
    """Status-code contract of include/gpry_hip.h (SURVEY.md 8b "Errors"): misuse comes back as a
    negative status plus a message from gpry_last_error, never as a crash, and a rejected call
    leaves the context usable."""
    import ctypes as C
    from gpry_amd import _lib
    lib = _lib.load_library()
    dev = _lib.Device(0)
    h = dev._h
    vp = C.c_void_p

    def err():
        return lib.gpry_last_error(h).decode()

    def ptr(a):
        return a.ctypes.data_as(vp)

    info, lml = C.c_int(0), C.c_double(0.0)
    # a NULL handle is refused by every entry point; destroy(NULL) is a no-op like free(NULL)
    assert lib.gpry_factorize(None, C.byref(info)) == -1
    assert b"ctx is NULL" in lib.gpry_last_error(None)
    assert lib.gpry_ctx_sync(None) == -1 and lib.gpry_ctx_destroy(None) == 0
    assert lib.gpry_comm_barrier(None) == -1 and lib.gpry_comm_destroy(None) == 0
    # calls out of order
    assert lib.gpry_factorize(h, C.byref(info)) == -1 and "gpry_set_train" in err()
    th = np.zeros(4)
    assert lib.gpry_set_theta(h, 3, ptr(th)) == -1 and "before set_train" in err()
    # bad sizes / pointers
    X = np.random.default_rng(0).uniform(0, 1, (40, 3))
    y, a = X.sum(1), np.full(40, 1e-6)
    assert lib.gpry_set_train(h, ptr(X), ptr(y), ptr(a), 0, 3) == -1
    assert lib.gpry_set_train(h, ptr(X), ptr(y), ptr(a), 40, 33) == -1 and "d=33" in err()
    assert lib.gpry_set_train(h, None, ptr(y), ptr(a), 40, 3) == -1 and "NULL" in err()
    assert lib.gpry_set_train(h, ptr(X), ptr(y), ptr(a), 40, 3) == 0
    assert lib.gpry_set_theta(h, 7, ptr(th)) == -1 and "kernel id" in err()
    bad = np.array([0.0, np.nan, 0.0, 0.0])
    assert lib.gpry_set_theta(h, 3, ptr(bad)) == -1 and "theta[1]" in err()
    assert lib.gpry_set_theta(h, 3, ptr(th)) == 0
    assert lib.gpry_lml(h, ptr(bad), 0, C.byref(lml), None, C.byref(info)) == -1 and "theta[1]" in err()
    assert lib.gpry_lml(h, ptr(th), 1, C.byref(lml), None, C.byref(info)) == -1 and "grad" in err()
    mean = np.zeros(5)
    assert lib.gpry_predict(h, ptr(X), 5, None, ptr(mean), None) == -1 and "not factorised" in err()
    assert lib.gpry_ctx_set_option(h, b"no_such_option", 1) == -1 and "unknown option" in err()
    # ... and the context still works after all the refusals (theta kept, not half-overwritten)
    assert lib.gpry_lml(h, ptr(th), 0, C.byref(lml), None, C.byref(info)) == 0 and info.value == 0
    ref = orc.log_marginal_likelihood(X, y, a, th, 3)
    assert abs(lml.value - ref) <= 1e-9 * abs(ref)
    assert lib.gpry_factorize(h, C.byref(info)) == 0 and info.value == 0
    assert lib.gpry_predict(h, ptr(X), 5, None, ptr(mean), None) == 0
    np.testing.assert_allclose(mean, y[:5], atol=1e-3)
    dev.close()


def test_two_contexts_driven_from_two_threads():
    """SURVEY.md 8b "Threading": a context is not re-entrant, but distinct contexts may be driven from
    distinct threads (ctypes releases the GIL).  Two models on one device, worked concurrently, give
    bit for bit what each gives alone."""
    import threading
    from gpry_amd import _lib

    def work(dev, seed, out):
        rng = np.random.default_rng(seed)
        N, d, M = 700 + 100 * seed, 4 + seed, 5000
        X = rng.uniform(0, 1, (N, d))
        y = np.sin(3 * X).sum(1)
        Xc = rng.uniform(0, 1, (M, d))
        theta = np.log(np.array([2.0] + [0.3 + 0.05 * k for k in range(d)]))
        res = []
        for _ in range(3):
            dev.set_train(X, y, np.full(N, 1e-6))
            dev.set_theta(3, theta)
            assert dev.factorize() == 0
            lml, grad, info = dev.lml(theta, True)
            sw = dev.sweep_logexp(Xc, 0.2, float(y.max()), 1e-3)
            top, bound = dev.sweep_topk(8)
            res.append((lml, grad.copy(), sw["acq"].copy(), top["idx"].copy(), bound))
        out[seed] = res

    devs = [_lib.Device(0), _lib.Device(0)]
    alone, together = {}, {}
    for s in (0, 1):
        work(devs[s], s, alone)
    threads = [threading.Thread(target=work, args=(devs[s], s, together)) for s in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for s in (0, 1):
        assert len(together[s]) == 3
        for (l0, g0, a0, i0, b0), (l1, g1, a1, i1, b1) in zip(alone[s], together[s]):
            assert l0 == l1 and b0 == b1
            np.testing.assert_array_equal(g0, g1)
            np.testing.assert_array_equal(a0, a1)
            np.testing.assert_array_equal(i0, i1)
    for dv in devs:
        dv.close()
