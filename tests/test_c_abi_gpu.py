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
