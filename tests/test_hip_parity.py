"""GPU parity tests: the HIP path (through the C ABI / the host mirror) against
(a) golden vectors captured from the real reference and (b) the pinned CPU oracle.

Tolerances (BASELINE.json north_star: posterior mean/variance within 1e-6 rel,
acquisition argmax identical): mean rel <= 1e-8, |d var| <= 1e-9 * C (variance compared
relative to the prior variance C), LML rel <= 1e-10, gradient
rel <= 1e-7 of its largest entry, kernel values rel <= 1e-13.
"""
import numpy as np
import pytest

from conftest import load_golden
from oracle import gpry_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from gpry_amd import _lib
    d = _lib.Device(0)
    yield d
    d.close()


def relmax(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / (np.max(np.abs(b)) + 1e-300))


# ---------------------------------------------------------------------------- MFMA engine
@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_mfma_gemm_layout_asymmetric(dev, at, bt):
    """A = I with an asymmetric B catches a transposed C/D fragment map."""
    n = 128
    B = np.arange(n * n, dtype=float).reshape(n, n) / 7.0
    A = np.eye(n)
    Ain = np.ascontiguousarray(A.T) if at else A
    Bin = np.ascontiguousarray(B.T) if bt else B
    np.testing.assert_array_equal(dev.debug_gemm(Ain, Bin, None, n, n, n, at, bt), B)
    rng = np.random.default_rng(1)
    A = rng.standard_normal((192, 320))
    B = rng.standard_normal((320, 256))
    Ain = np.ascontiguousarray(A.T) if at else A
    Bin = np.ascontiguousarray(B.T) if bt else B
    assert relmax(dev.debug_gemm(Ain, Bin, None, 192, 256, 320, at, bt), A @ B) < 1e-14


@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 0)])
def test_small_tile_gemm_gives_the_bits_of_the_128_tile_engines(dev, at, bt):
    """Launches with a handful of tiles run with 64 x 64 tiles (gemm_small.hip: the V = L^-1 levels and K^-1 of a few
    hundred training points): every layout, triangular k-range, epilogue and the tile-level lower_only rule against
    numpy and, bit for bit, against the 128 x 128 engines."""
    rng = np.random.default_rng(11)
    SMALL = 1 << 28
    M, N, K = 192, 320, 256
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N)); C0 = rng.standard_normal((M, N))
    Ain = np.ascontiguousarray(A.T) if at else A
    Bin = np.ascontiguousarray(B.T) if bt else B
    try:
        dev.set_option("gemm_small", 1000)
        for epi, ref in ((0, A @ B), (1, -(A @ B)), (2, C0 - A @ B)):
            got = dev.debug_gemm(Ain, Bin, C0 if epi == 2 else None, M, N, K, at, bt, epi=epi, tile_map=SMALL)
            big = dev.debug_gemm(Ain, Bin, C0 if epi == 2 else None, M, N, K, at, bt, epi=epi)
            assert relmax(got, ref) < 1e-14
            np.testing.assert_array_equal(got, big)
        # triangular operands: square, the k-range of a tile is cut at the tile's own origin
        n = 384
        L = np.tril(rng.standard_normal((n, n))); G = rng.standard_normal((n, n))
        cases = []
        if not at and not bt:
            cases = [(1, L, G, L @ G, 0), (2, G, L, G @ L, 0)]                # KM_A_LOWER, KM_B_LOWER
        if at and not bt:
            cases = [(3, L, L, np.tril(L.T @ L), 1)]                           # KM_AT_LOWER_B_LOWER, lower tiles only
        for kmode, P, Q, ref, lower in cases:
            got = dev.debug_gemm(P, Q, None, n, n, n, at, bt, kmode=kmode, lower_only=lower, tile_map=SMALL)
            big = dev.debug_gemm(P, Q, None, n, n, n, at, bt, kmode=kmode, lower_only=lower)
            if lower:
                blk = np.kron(np.tril(np.ones((n // 64, n // 64))), np.ones((64, 64))) > 0     # tiles of the small engine
                full = L.T @ L
                assert relmax(got[blk], full[blk]) < 1e-13
                got, big = np.tril(got), np.tril(big)
            else:
                assert relmax(got, ref) < 1e-13
            np.testing.assert_array_equal(got, big)
    finally:
        dev.set_option("gemm_small", 32)


def test_mfma_gemm_triangular_modes_and_sumsq(dev):
    rng = np.random.default_rng(2)
    n = 512
    L = np.tril(rng.standard_normal((n, n)))
    B = rng.standard_normal((n, 640))
    ref = L @ B
    ss = np.stack([np.sum(ref[t * 128:(t + 1) * 128] ** 2, axis=0) for t in range(n // 128)])
    for tile_map in (0, 1):
        got = dev.debug_gemm(L, B, None, n, 640, n, kmode=1, epi=3, tile_map=tile_map)
        assert relmax(got, ss) < 1e-14
    C0 = rng.standard_normal((n, n))
    got = dev.debug_gemm(L, L, C0, n, n, n, b_trans=1, epi=2)   # C -= L L^T
    assert relmax(got, C0 - L @ L.T) < 1e-13


@pytest.mark.parametrize("at,bt", [(0, 0), (0, 1), (1, 0)])
def test_dma_pipelined_gemm_matches_numpy_and_the_register_staged_engine(dev, at, bt):
    """gemm_dma.hip (128-aligned products of the factor chain) in every layout it builds, every
    epilogue, the triangular k-ranges, lower-only tiles and split-K; `gemm_dma=0` routes the same
    call through the register-staged kernel."""
    rng = np.random.default_rng(20 + 2 * at + bt)
    M, N, K = 384, 256, 512
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N))
    Ain = np.ascontiguousarray(A.T) if at else A
    Bin = np.ascontiguousarray(B.T) if bt else B
    C0 = rng.standard_normal((M, N))
    for epi, ref in ((0, A @ B), (1, -(A @ B)), (2, C0 - A @ B)):
        outs = []
        for dma in (1, 0):
            dev.set_option("gemm_dma", dma)
            outs.append(dev.debug_gemm(Ain, Bin, C0 if epi == 2 else None, M, N, K, at, bt, epi=epi))
        dev.set_option("gemm_dma", 1)
        assert relmax(outs[0], ref) < 1e-14
        np.testing.assert_array_equal(outs[0], outs[1])      # same accumulation order
    # split-K (test hook: bits 8..11 of tile_map), 2 and 4 slices
    for ns in (2, 4):
        got = dev.debug_gemm(Ain, Bin, None, M, N, K, at, bt, epi=1, tile_map=ns << 8)
        assert relmax(got, -(A @ B)) < 1e-14
    # triangular operands: only the k-range that holds non-zeros is walked
    n = 512
    L = np.tril(rng.standard_normal((n, n)))
    G = rng.standard_normal((n, n))
    if (at, bt) == (0, 0):
        assert relmax(dev.debug_gemm(L, G, None, n, n, n, kmode=1), L @ G) < 1e-14          # A lower
        assert relmax(dev.debug_gemm(G, L, None, n, n, n, kmode=2), G @ L) < 1e-14          # B lower
        assert relmax(dev.debug_gemm(G, L.T.copy(), None, n, n, n, kmode=4), G @ L.T) < 1e-14   # B upper
    if (at, bt) == (1, 0):
        got = dev.debug_gemm(L, L, None, n, n, n, a_trans=1, kmode=3, lower_only=True)    # lauum: V^T V
        assert relmax(np.tril(got), np.tril(L.T @ L)) < 1e-14
        assert np.all(np.triu(got[:128, 128:]) == 0.0)          # upper tiles are not touched
    if (at, bt) == (0, 1):
        C1 = rng.standard_normal((n, n))
        got = dev.debug_gemm(L, L, C1, n, n, n, b_trans=1, epi=2, lower_only=True)        # SYRK update
        ref = C1 - L @ L.T
        assert relmax(np.tril(got), np.tril(ref)) < 1e-13
        np.testing.assert_array_equal(got[:128, 128:], C1[:128, 128:])


# ---------------------------------------------------------------------------- F1 kernels
@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("d", [1, 2, 5])
def test_f1_kernel_matrices_vs_reference(dev, kid, d):
    g = load_golden("kernels")
    p = f"f1_k{kid}_d{d}_"
    X, Y, theta = g[p + "X"], g[p + "Y"], g[p + "theta"]
    dev.set_train(X, np.zeros(len(X)), np.zeros(len(X)))
    dev.set_theta(kid, theta)
    K = dev.kernel_train(add_alpha=False)
    np.testing.assert_allclose(K, g[p + "K"], rtol=1e-13, atol=0)
    assert np.array_equal(K, K.T)
    np.testing.assert_allclose(dev.kernel_cross(Y), g[p + "Kx"], rtol=1e-13, atol=0)
    Ka = dev.kernel_train(add_alpha=True)
    np.testing.assert_array_equal(Ka, K)  # alpha = 0 here


# ---------------------------------------------------------------------------- F2 / F3
@pytest.mark.parametrize("kid", [0, 1, 2, 3])
def test_f2_factor_and_f3_lml_vs_reference(dev, kid):
    g = load_golden("factor_lml")
    p = f"f2_k{kid}_"
    dev.set_train(g[p + "X_"], g[p + "y_"], g[p + "alpha"])
    dev.set_theta(kid, g[p + "theta"])
    assert dev.factorize() == 0
    L, V, a = dev.get_factor()
    assert not np.triu(L, 1).any() and not np.triu(V, 1).any()
    np.testing.assert_allclose(L, g[p + "L"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(V, g[p + "V"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(a, g[p + "alpha_"], rtol=1e-8, atol=1e-9)
    th = g[f"f3_k{kid}_theta"]
    lml, grad, info = dev.lml(th, True)
    assert info == 0
    assert abs(lml - g[f"f3_k{kid}_lml"]) <= 1e-10 * abs(g[f"f3_k{kid}_lml"])
    assert relmax(grad, g[f"f3_k{kid}_grad"]) < 1e-7
    lml2, _ = dev.lml(th, False)
    assert lml2 == lml
    # the prediction factor is untouched by lml()
    L2, _, _ = dev.get_factor(want_V=False, want_alpha=False)
    np.testing.assert_array_equal(L, L2)


def test_f3_non_positive_definite_convention(dev):
    g = load_golden("factor_lml")
    X_, y_, th = g["f3_nonpd_X_"], g["f3_nonpd_y_"], g["f3_nonpd_theta"]
    dev.set_train(X_, y_, np.zeros(len(y_)))
    lml, grad, info = dev.lml(th, True)
    assert lml == -np.inf and not grad.any() and info > 0   # sklearn:_gpr.py:586-589
    dev.set_theta(0, th)
    assert dev.factorize() > 0


# ---------------------------------------------------------------------------- F4 / F5 / F8
def _oracle_model(g, p, kid, **kw):
    m = orc.OracleGPR(g[p + "bounds"], kernel_id=kid, **kw)
    m.theta = np.array(g[p + "theta"])
    m.fitted = True
    return m


def _load_model(dev, m):
    dev.set_train(m.X_train_, m.y_train_, m.alpha)
    dev.set_theta(m.kernel_id, m.theta)
    dev.set_affine(m.pre_X.lo, m.pre_X.hi - m.pre_X.lo, m.pre_y.mean_, m.pre_y.std_, m.clip_hi())
    assert dev.factorize() == 0


@pytest.mark.parametrize("kid", [0, 3])
def test_f4_predict_clip_trust_vs_reference(dev, kid):
    from gpry_amd import _lib
    g = load_golden("predict")
    p = f"f4_k{kid}_"
    m = _oracle_model(g, p, kid, clip_factor=1.0)
    m.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    Xc = g[p + "Xc"]
    mean, std = dev.predict(Xc, return_std=True)
    assert (mean == g[p + "clip_hi"]).sum() == (g[p + "mean"] == g[p + "clip_hi"]).sum() >= 1
    np.testing.assert_allclose(mean, g[p + "mean"], rtol=1e-8, atol=1e-9)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    assert np.max(np.abs(std ** 2 - g[p + "std"] ** 2)) <= 1e-9 * C
    # mean-only calls take the fused low-latency path (other summation order)
    np.testing.assert_allclose(dev.predict(Xc), mean, rtol=1e-9, atol=1e-10)
    outside = ~orc.is_in_bounds(Xc, g[p + "trust_bounds"])
    mask = outside.astype(np.uint8) * _lib.MASK_OUTSIDE_TRUST
    mean_tr, std_tr = dev.predict(Xc, return_std=True, mask=mask)
    assert np.array_equal(np.isneginf(mean_tr), np.isneginf(g[p + "mean_tr"]))
    fin = np.isfinite(mean_tr)
    np.testing.assert_allclose(mean_tr[fin], g[p + "mean_tr"][fin], rtol=1e-8, atol=1e-9)
    assert np.max(np.abs(std_tr ** 2 - g[p + "std_tr"] ** 2)) <= 1e-9 * C


def test_f4_classifier_mask_vs_reference(dev):
    from gpry_amd import _lib
    g = load_golden("predict")
    p = "f4_svm_"
    m = _oracle_model(g, p, 3)
    m.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    mask = (~g[p + "finite"]).astype(np.uint8) * _lib.MASK_CLASSIFIED_INF
    mean, std = dev.predict(g[p + "Xc"], return_std=True, mask=mask)
    assert np.array_equal(np.isneginf(mean), np.isneginf(g[p + "mean"]))
    fin = g[p + "finite"]
    np.testing.assert_allclose(mean[fin], g[p + "mean"][fin], rtol=1e-8, atol=1e-9)
    assert not std[~fin].any()
    np.testing.assert_allclose(std, g[p + "std"], rtol=1e-6, atol=1e-9)


def test_f5_logexp_through_the_sweep(dev):
    """Drive the fused epilogue with a model whose (mean, std) are known, then compare the
    acquisition with LogExp.f of the reference on those very values."""
    g = load_golden("predict")
    p = "f4_k3_"
    m = _oracle_model(g, p, 3, clip_factor=1.0)
    m.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    zeta, base, noise = float(g["f5_zeta"]), float(g["f5_baseline"]), float(g["f5_noise"])
    out = dev.sweep_logexp(g[p + "Xc"], zeta, base, noise)
    ref = orc.logexp_f(out["y"], out["sigma"], base, noise, zeta)
    assert out["n_nan"] == 0
    assert np.array_equal(np.isneginf(out["acq"]), np.isneginf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_allclose(out["acq"][fin], ref[fin], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["y"], g[p + "mean"], rtol=1e-8, atol=1e-9)


def test_f8_append_rows_fixed_theta_vs_reference(dev):
    g = load_golden("predict")
    m = _oracle_model(g, "f8_", 2)
    X, y, Xc = g["f8_X"], g["f8_y"], g["f8_Xc"]
    m.append_to_data(X[:32], y[:32], fit_gpr=False, fit_preprocessors=True)
    m.append_to_data(X[32:], y[32:], fit_gpr=False, fit_preprocessors=False)
    _load_model(dev, m)
    L, V, a = dev.get_factor()
    np.testing.assert_allclose(L, g["f8_L"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(V, g["f8_V"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(a, g["f8_alpha_"], rtol=1e-8, atol=1e-9)
    mean, std = dev.predict(Xc, return_std=True)
    np.testing.assert_allclose(mean, g["f8_mean_after"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(std, g["f8_std_after"], rtol=1e-6, atol=1e-9)


# ---------------------------------------------------------------------------- oracle at size
@pytest.mark.parametrize("N,d,kid,M", [(1, 1, 0, 1), (2, 3, 3, 5), (127, 4, 1, 129), (129, 2, 2, 127),
                                       (1024, 8, 0, 2000), (1500, 32, 3, 300)])
def test_pipeline_vs_oracle_ragged_sizes(dev, N, d, kid, M):
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=N + d)
    m = orc.OracleGPR(bounds, kernel_id=kid, normalize_y=(N > 1))
    m.theta = np.log(np.array([4.0] + [0.3 + 0.01 * k for k in range(d)]))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    rm, rs = m.predict(Xc, return_std=True)
    mean, std = dev.predict(Xc, return_std=True)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    np.testing.assert_allclose(mean, rm, rtol=1e-8, atol=1e-8)
    assert np.max(np.abs(std ** 2 - rs ** 2)) <= 1e-9 * C
    th = m.theta + 0.05
    lml, grad, info = dev.lml(th, True)
    rl, rg = orc.log_marginal_likelihood(m.X_train_, m.y_train_, m.alpha, th, kid, True)
    assert info == 0 and abs(lml - rl) <= 1e-10 * max(1.0, abs(rl))
    assert np.max(np.abs(grad - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg)))
    zeta = orc.auto_zeta(d)
    out = dev.sweep_logexp(Xc, zeta, m.y_max, m.noise_level)
    racq = orc.logexp_f(rm, rs, m.y_max, m.noise_level, zeta)
    if np.isfinite(racq).any():
        assert int(np.argmax(out["acq"])) == int(np.argmax(racq))   # acquisition argmax identical
    K = min(M, 50)
    top, bound = dev.sweep_topk(K)
    order = np.lexsort((-np.arange(M), -out["acq"]))
    np.testing.assert_array_equal(top["idx"], order[:K])
    np.testing.assert_array_equal(top["acq"], out["acq"][order[:K]])
    assert bound == (out["acq"][order[K]] if K < M else -np.inf)


def test_topk_ties_exclusions_and_exhaustion(dev):
    bounds, X, y, Xc = orc.synthetic_like_goldens(64, 2, 1000, seed=5)
    Xc[100:110] = Xc[5]          # duplicated candidates -> tied acquisition values
    m = orc.OracleGPR(bounds, kernel_id=0)
    m.theta = np.log(np.array([4.0, 0.3, 0.3]))
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    out = dev.sweep_logexp(Xc, 0.5, m.y_max, m.noise_level)
    acq = out["acq"]
    assert len(np.unique(acq[100:110])) == 1
    order = np.lexsort((-np.arange(1000), -acq))   # (acq desc, idx desc)
    for K in (1, 7, 400, 1000, 5000):
        top, bound = dev.sweep_topk(K)
        k = min(K, 1000)
        np.testing.assert_array_equal(top["idx"], order[:k])
    excl = order[:13]
    top, bound = dev.sweep_topk(20, exclude=excl)
    np.testing.assert_array_equal(top["idx"], order[13:33])
    assert bound == acq[order[33]]
    # re-using the resident candidate set (X=None) gives the same sweep
    out2 = dev.sweep_logexp(None, 0.5, m.y_max, m.noise_level, M=1000)
    np.testing.assert_array_equal(out2["acq"], acq)


@pytest.mark.parametrize("N,d,kid", [(1024, 8, 0), (4096, 16, 3), (8192, 20, 3)])
def test_full_size_properties(dev, N, d, kid):
    """BASELINE configs 2, 3 and 5 sizes: size-independent properties of the device path."""
    M = 20000
    bounds, X, y, Xc = orc.synthetic_problem(N, d, M)
    pre = orc.NormalizeBounds(bounds)
    X_ = pre.transform(X)
    ym, ys = y.mean(), y.std()
    alpha = np.full(N, (1e-2 / ys) ** 2)
    theta = np.log(np.array([4.0] + [0.3] * d))
    dev.set_train(X_, (y - ym) / ys, alpha)
    dev.set_theta(kid, theta)
    dev.set_affine(pre.lo, pre.hi - pre.lo, ym, ys, np.inf)
    assert dev.factorize() == 0
    L, V, a = dev.get_factor()
    K = dev.kernel_train(add_alpha=True)
    assert np.array_equal(K, K.T)
    assert relmax(L @ L.T, K) < 1e-13                      # factor reproduces K + alpha I
    assert np.max(np.abs(V @ L - np.eye(N))) < 1e-9         # V is the inverse factor
    assert relmax(K @ a, (y - ym) / ys) < 1e-8              # alpha_ solves the system
    # interpolation: at the training points the posterior mean is y up to the noise
    mean, std = dev.predict(X[:512], return_std=True)
    assert np.max(np.abs(mean - y[:512])) < 0.1
    assert np.all(std < 0.1 * np.sqrt(np.exp(theta[0])) * ys)
    # variance identity on a sample: var = C - |V k*|^2 with the exported V
    Kx = dev.kernel_cross(pre.transform(Xc[:256]))
    var_ref = np.exp(theta[0]) - np.sum((V @ Kx.T) ** 2, axis=0)
    _, s = dev.predict(Xc[:256], return_std=True)
    assert np.max(np.abs((s / ys) ** 2 - np.clip(var_ref, 0, None))) < 1e-10 * np.exp(theta[0])
    # LML gradient against central finite differences of the device LML itself
    lml, grad, info = dev.lml(theta, True)
    for k in (0, 1, d):
        e = np.zeros(d + 1)
        e[k] = 1e-5
        fd = (dev.lml(theta + e, False)[0] - dev.lml(theta - e, False)[0]) / 2e-5
        assert abs(fd - grad[k]) <= 1e-5 * max(1.0, abs(grad[k]))


@pytest.mark.parametrize("N", [1100, 1250, 1500, 1600, 2100, 2400, 3100])
def test_factor_chain_at_ragged_block_counts(dev, N):
    """Padded sizes that are not powers of two (Np = 1152 ... 3200): the last outer block of the Cholesky
    is narrower than the others, the V = L^-1 tree has ragged nodes (levels that mix the DMA-pipelined
    and the register-staged GEMM), split-K factors change from level to level.  Size-independent
    properties, and the gradient traces (which read K^-1) against finite differences of the LML."""
    d = 5
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d))
    y = np.sin(4 * X).sum(1) + 0.1 * rng.standard_normal(N)
    alpha = np.full(N, 1e-4)
    theta = np.log(np.array([3.0, 0.3, 0.4, 0.5, 0.6, 0.7]))
    dev.set_train(X, y, alpha)
    dev.set_theta(3, theta)
    assert dev.factorize() == 0
    L, V, a = dev.get_factor()
    K = dev.kernel_train(add_alpha=True)
    assert relmax(L @ L.T, K) < 1e-13
    assert np.max(np.abs(V @ L - np.eye(N))) < 1e-8
    assert np.all(np.triu(V, 1) == 0.0) and np.all(np.triu(L, 1) == 0.0)
    assert relmax(K @ a, y) < 1e-7
    lml, grad, info = dev.lml(theta, True)
    assert info == 0
    sign, logdet = np.linalg.slogdet(K)
    ref = -0.5 * y @ a - 0.5 * logdet - 0.5 * N * np.log(2 * np.pi)
    assert abs(lml - ref) <= 1e-10 * abs(ref)
    for k in (0, 2, d):
        e = np.zeros(d + 1)
        e[k] = 1e-5
        fd = (dev.lml(theta + e, False)[0] - dev.lml(theta - e, False)[0]) / 2e-5
        assert abs(fd - grad[k]) <= 2e-5 * max(1.0, abs(grad[k]))
    # the register-staged engine gives the same factor bit for bit (same accumulation order)
    dev.set_option("gemm_dma", 0)
    assert dev.factorize() == 0
    L0, V0, _ = dev.get_factor()
    dev.set_option("gemm_dma", 1)
    assert np.array_equal(L0, L) and np.array_equal(V0, V)


def test_rccl_communicator_single_rank(dev):
    """RCCL entry points on a 1-rank communicator (the multi-rank merge logic is covered by
    tests/test_multirank_cpu.py; 8-GPU runs are the driver's)."""
    from gpry_amd import _lib
    comm = _lib.RcclComm(dev, 1, 0, _lib.RcclComm.unique_id())
    rec = np.zeros(5, dtype=_lib.CAND_DTYPE)
    rec["acq"] = np.arange(5.0)
    rec["idx"] = np.arange(5) * 7
    got = comm.allgather(rec)
    assert got.shape == (1, 5) and np.array_equal(got[0], rec)
    x = np.array([3.0, -np.inf, 7.5])
    np.testing.assert_array_equal(comm.allreduce_max(x), x)
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("N,d,kid", [(60, 2, 0), (300, 5, 3), (1100, 7, 1), (2500, 3, 2)])
def test_one_point_call_equals_predict_plus_predict_grad(N, d, kid):
    """gpry_predict_point (mean, std and both x-gradient contractions of ONE point in one call: the acquisition
    optimiser's step) against gpry_predict + gpry_predict_grad, with an X map, clipping and the two mask bits."""
    from gpry_amd import _lib
    rng = np.random.default_rng(N)
    X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1)
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, np.full(N, 1e-6)); dv.set_theta(kid, np.log(np.array([2.0] + [0.4] * d)))
        lo, span = -0.5 * np.ones(d), 2.0 * np.ones(d)
        dv.set_affine(lo, span, 0.3, 1.7, 1.2)
        assert dv.factorize() == 0
        for trial in range(6):
            x = lo + span * rng.uniform(size=d)
            if trial == 5:
                x = lo + span * X[3]                                  # on a training point: std ~ 0
            m0, s0 = dv.predict(x[None, :], return_std=True)
            mg0, kg0 = dv.predict_grad(x)
            m1, s1, mg1, kg1, bits1 = dv.predict_point(x)
            assert bits1 == 0
            C = 2.0 * 1.7 ** 2
            assert abs(m1 - m0[0]) <= 1e-10 * max(1.0, abs(m0[0]))
            assert abs(s1 ** 2 - s0[0] ** 2) <= 1e-10 * C
            np.testing.assert_allclose(mg1, mg0, rtol=1e-10, atol=1e-11)
            np.testing.assert_allclose(kg1, kg0, rtol=1e-9, atol=1e-10 * max(1.0, np.max(np.abs(kg0))))
            _, _, mg2, kg2, _ = dv.predict_point(x, want_kinv=False)
            np.testing.assert_array_equal(mg2, mg1)
            assert np.all(kg2 == 0.0)
        for bits, exp_std in ((_lib.MASK_OUTSIDE_TRUST, None), (_lib.MASK_CLASSIFIED_INF, 0.0)):
            m1, s1, _, _, vb = dv.predict_point(x, mask_bits=bits)
            assert vb == bits
            mk = np.array([bits], dtype=np.uint8)
            m0, s0 = dv.predict(x[None, :], return_std=True, mask=mk)
            assert np.isneginf(m1) and np.isneginf(m0[0])
            assert s1 == (exp_std if exp_std is not None else s1) and abs(s1 - s0[0]) <= 1e-12 * max(1.0, s0[0])
        # the device's own gates (option predict_gates): a trust box that leaves the last point outside
        dv.set_option("predict_gates", 1)
        box = np.column_stack([x - 2.0, x - 1.0])
        dv.set_gates(trust_bounds=box)
        m1, s1, _, _, vb = dv.predict_point(x)
        assert vb == _lib.MASK_OUTSIDE_TRUST and np.isneginf(m1) and s1 > 0.0
        dv.set_gates(trust_bounds=np.column_stack([x - 1.0, x + 1.0]))
        m1, s1, _, _, vb = dv.predict_point(x)
        assert vb == 0 and np.isfinite(m1)
    finally:
        dv.close()


def test_predictions_after_an_objective_evaluation_at_another_theta():
    """gpry_lml scales the training coordinates for ITS theta and no longer puts the prediction factor's back itself
    (one dispatch per evaluation of every fit): every reader on behalf of the prediction factor restores them first --
    the panel build of predict / sweep, the one-launch and the resident mean paths, x-gradients, bordered appends."""
    from gpry_amd import _lib
    rng = np.random.default_rng(12)
    N, d = 300, 4
    X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1)
    th1 = np.log(np.array([2.0, 0.3, 0.4, 0.5, 0.35])); th2 = th1 + 0.3
    Xc = rng.uniform(size=(50, d))
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, np.full(N, 1e-6)); dv.set_theta(3, th1)
        assert dv.factorize() == 0
        ref_ms = dv.predict(Xc, return_std=True)
        ref_m = dv.predict(Xc)
        ref_1 = [dv.predict(Xc[i:i + 1])[0] for i in range(5)]            # resident kernel
        dv.set_option("predict_serve", 0)
        ref_1l = [dv.predict(Xc[i:i + 1])[0] for i in range(5)]           # one launch per call
        dv.set_option("predict_serve", 1)
        ref_g = dv.predict_grad(Xc[0]) + dv.predict_grad_batch(Xc[:7])
        for reader in ("std", "mean", "serve", "launch", "grad", "append"):
            dv.lml(th2, True)                                             # leaves its own scaled coordinates behind
            if reader == "std":
                got = dv.predict(Xc, return_std=True)
                assert np.array_equal(got[0], ref_ms[0]) and np.array_equal(got[1], ref_ms[1])
            elif reader == "mean":
                assert np.array_equal(dv.predict(Xc), ref_m)
            elif reader == "serve":
                assert [dv.predict(Xc[i:i + 1])[0] for i in range(5)] == ref_1
            elif reader == "launch":
                dv.set_option("predict_serve", 0)
                assert [dv.predict(Xc[i:i + 1])[0] for i in range(5)] == ref_1l
                dv.set_option("predict_serve", 1)
            elif reader == "grad":
                got = dv.predict_grad(Xc[0])
                dv.lml(th2, True)
                got = got + dv.predict_grad_batch(Xc[:7])
                for a, b in zip(got, ref_g):
                    assert np.array_equal(np.asarray(a), np.asarray(b))
            elif reader == "append":
                xn = rng.uniform(size=(3, d)); yn = np.sin(3 * xn).sum(1)
                assert dv.append_rows(xn, yn, 1e-6) == 0
                fresh = _lib.Device(0)
                try:
                    fresh.set_train(np.vstack([X, xn]), np.concatenate([y, yn]), np.full(N + 3, 1e-6)); fresh.set_theta(3, th1)
                    assert fresh.factorize() == 0
                    np.testing.assert_allclose(dv.predict(Xc), fresh.predict(Xc), rtol=1e-9, atol=1e-10)
                finally:
                    fresh.close()
    finally:
        dv.close()


def test_an_x_map_of_another_dimension_is_refused():
    """A gpry_set_affine made for a model of fewer dimensions has zero spans in the new ones: the prediction entry points
    say so instead of returning NaN."""
    from gpry_amd import _lib
    d2 = _lib.Device(0)
    try:
        rng = np.random.default_rng(0)
        d2.set_train(rng.uniform(size=(20, 2)), rng.standard_normal(20), np.full(20, 1e-6))
        d2.set_theta(0, np.zeros(3))
        d2.set_affine(np.zeros(2), np.ones(2), 0.0, 1.0, np.inf)
        assert d2.factorize() == 0
        assert np.all(np.isfinite(d2.predict(rng.uniform(size=(5, 2)), return_std=True)[1]))
        d2.set_train(rng.uniform(size=(20, 3)), rng.standard_normal(20), np.full(20, 1e-6))
        d2.set_theta(0, np.zeros(4))
        assert d2.factorize() == 0
        with pytest.raises(RuntimeError, match="affine map of X"):
            d2.predict(rng.uniform(size=(5, 3)), return_std=True)
        d2.set_affine(np.zeros(3), np.ones(3), 0.0, 1.0, np.inf)
        assert np.all(np.isfinite(d2.predict(rng.uniform(size=(5, 3)), return_std=True)[1]))
    finally:
        d2.close()


def test_factorize_adopts_the_factor_of_the_last_lml_evaluation(dev):
    """gpry_factorize at the theta of the preceding gpry_lml call swaps that evaluation's factor
    in instead of factorising again: same bits as a fresh factorisation, and no stale reuse once
    theta, the training set or the scratch matrices have changed."""
    bounds, X, y, Xc = orc.synthetic_like_goldens(300, 5, 64, seed=8)
    pre = orc.NormalizeBounds(bounds)
    X_ = pre.transform(X)
    alpha = np.full(300, 1e-4)
    th1 = np.log(np.array([3.0, 0.3, 0.4, 0.5, 0.35, 0.45]))
    th2 = th1 + 0.1
    dev.set_train(X_, y, alpha)
    dev.set_affine(np.zeros(5), np.ones(5), 0.0, 1.0, np.inf)     # (not the map a test of another dimension left behind)
    dev.set_option("lml_cache", 0)
    dev.set_theta(3, th1)
    assert dev.factorize() == 0
    L_ref, V_ref, a_ref = dev.get_factor()
    m_ref = dev.predict(pre.transform(Xc), return_std=True)
    dev.set_option("lml_cache", 1)
    dev.timing_reset()
    dev.lml(th2, True)
    dev.lml(th1, True)                       # last evaluation at th1
    n_potrf = dev.timing("potrf")[1]
    dev.set_theta(3, th1)
    assert dev.factorize() == 0
    assert dev.timing("potrf")[1] == n_potrf            # adopted, not refactorised
    L, V, a = dev.get_factor()
    assert np.array_equal(L, L_ref) and np.array_equal(V, V_ref) and np.array_equal(a, a_ref)
    m = dev.predict(pre.transform(Xc), return_std=True)
    assert np.array_equal(m[0], m_ref[0]) and np.array_equal(m[1], m_ref[1])
    # a different theta, or an evaluation in between that reused the scratch matrices: refactorise
    dev.lml(th1, True)
    dev.set_theta(3, th2)
    assert dev.factorize() == 0 and dev.timing("potrf")[1] == n_potrf + 2
    dev.lml(th2, False)
    dev.kernel_train()
    assert dev.factorize() == 0 and dev.timing("potrf")[1] == n_potrf + 4
    ref2 = orc.log_marginal_likelihood(X_, y, alpha, th2, 3)
    assert abs(dev.lml(th2, False)[0] - ref2) <= 1e-10 * abs(ref2)


@pytest.mark.parametrize("M,chunk", [(70001, 32768), (5003, 1024)])
def test_multi_chunk_sweep_with_masks_vs_oracle(dev, M, chunk):
    """Chunked sweep (ragged last chunk), per-candidate masks, resident re-use and the shortlist
    against the oracle evaluated in one piece."""
    from gpry_amd import _lib
    bounds, X, y, Xc = orc.synthetic_like_goldens(300, 6, M, seed=17)
    m = orc.OracleGPR(bounds, kernel_id=2)
    m.theta = np.log(np.array([4.0] + [0.3] * 6))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    rng = np.random.default_rng(0)
    mask = np.zeros(M, dtype=np.uint8)
    mask[rng.random(M) < 0.05] = _lib.MASK_CLASSIFIED_INF
    mask[rng.random(M) < 0.05] |= _lib.MASK_OUTSIDE_TRUST
    rm, rs = m.predict(Xc, return_std=True)
    rm = rm.copy(); rs = rs.copy()
    rm[mask != 0] = -np.inf
    rs[(mask & _lib.MASK_CLASSIFIED_INF) != 0] = 0.0
    zeta = orc.auto_zeta(6)
    racq = orc.logexp_f(rm, rs, m.y_max, m.noise_level, zeta)
    try:
        dev.set_option("sweep_chunk", chunk)
        out = dev.sweep_logexp(Xc, zeta, m.y_max, m.noise_level, mask=mask)
    finally:
        dev.set_option("sweep_chunk", 0)
    fin = np.isfinite(rm)
    assert np.array_equal(np.isneginf(out["y"]), ~fin)
    np.testing.assert_allclose(out["y"][fin], rm[fin], rtol=1e-8, atol=1e-8)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    assert np.max(np.abs(out["sigma"] ** 2 - rs ** 2)) <= 1e-9 * C
    assert np.array_equal(np.isneginf(out["acq"]), np.isneginf(racq))
    ok = np.isfinite(racq)
    np.testing.assert_allclose(out["acq"][ok], racq[ok], rtol=1e-6, atol=1e-6)
    assert int(np.argmax(out["acq"])) == int(np.argmax(racq))
    top, bound = dev.sweep_topk(200)
    order = np.lexsort((-np.arange(M), -out["acq"]))
    np.testing.assert_array_equal(top["idx"], order[:200])
    assert bound == out["acq"][order[200]]


@pytest.mark.parametrize("M,chunk", [(70001, 32768), (5003, 1024), (900, 0)])
def test_pool_uploaded_underneath_the_sweep_gives_the_bits_of_the_upload_in_front(dev, M, chunk):
    """A fresh candidate pool (gpry/gp_acquisition.py:1023-1031: every mc_every-th call draws one) goes up chunk by chunk
    on the copy stream, chunk c + 1 underneath the kernels of chunk c (option "sweep_upload", default): y, sigma, acq and
    the shortlist are, bit for bit, those of ONE copy in front of the sweep -- with a host mask, ragged last chunk, a pool
    smaller than a chunk, and the resident re-use behind it; a second pool in the same buffers replaces the first."""
    from gpry_amd import _lib
    bounds, X, y, Xc = orc.synthetic_like_goldens(300, 6, M, seed=23)
    m = orc.OracleGPR(bounds, kernel_id=3)
    m.theta = np.log(np.array([4.0] + [0.3] * 6))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    rng = np.random.default_rng(1)
    mask = np.zeros(M, dtype=np.uint8)
    mask[rng.random(M) < 0.05] = _lib.MASK_CLASSIFIED_INF
    zeta = orc.auto_zeta(6)
    Xc2 = Xc[::-1].copy()
    res = {}
    try:
        dev.set_option("sweep_chunk", chunk)
        for mode in (0, 1):
            dev.set_option("sweep_upload", mode)
            a = dev.sweep_logexp(Xc, zeta, m.y_max, m.noise_level, mask=mask)
            top_a = dev.sweep_topk(100)
            b = dev.sweep_logexp(Xc2, zeta, m.y_max, m.noise_level)                 # another pool, no mask
            c = dev.sweep_logexp(None, zeta, m.y_max, m.noise_level, M=M)           # ... which is resident now
            res[mode] = (a, top_a, b, c)
    finally:
        dev.set_option("sweep_chunk", 0)
        dev.set_option("sweep_upload", 1)
    for k in ("y", "sigma", "acq"):
        np.testing.assert_array_equal(res[1][0][k], res[0][0][k])
        np.testing.assert_array_equal(res[1][2][k], res[0][2][k])
        np.testing.assert_array_equal(res[1][3][k], res[1][2][k])
    np.testing.assert_array_equal(res[1][1][0], res[0][1][0])
    assert res[1][1][1] == res[0][1][1]
    # the reversed pool gives the reversed arrays of the first one without its mask
    free = mask == 0
    np.testing.assert_array_equal(res[1][2]["acq"][::-1][free], res[1][0]["acq"][free])


@pytest.mark.parametrize("M", [1, 3, 16, 17])
def test_small_batch_paths_agree_with_the_panel_path(dev, M):
    """gpry_predict takes latency paths for small batches (fused mean kernel; k* rows + multi-vector
    triangular product for <= 16 points with std): same numbers as the panel path, masks included."""
    from gpry_amd import _lib
    bounds, X, y, Xc = orc.synthetic_like_goldens(1000, 7, M, seed=23)
    m = orc.OracleGPR(bounds, kernel_id=3)
    m.theta = np.log(np.array([4.0] + [0.3] * 7))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    mask = np.zeros(M, dtype=np.uint8)
    if M > 2:
        mask[1], mask[2] = _lib.MASK_CLASSIFIED_INF, _lib.MASK_OUTSIDE_TRUST
    try:
        dev.set_option("predict_small", 0)
        mean_p, std_p = dev.predict(Xc, return_std=True, mask=mask)
        dev.set_option("predict_small", 2048)
        mean_s, std_s = dev.predict(Xc, return_std=True, mask=mask)
        mean_only = dev.predict(Xc, mask=mask)
    finally:
        dev.set_option("predict_small", 2048)
    rm, rs = m.predict(Xc, return_std=True)
    fin = mask == 0
    np.testing.assert_allclose(mean_p[fin], rm[fin], rtol=1e-8, atol=1e-8)
    for got in (mean_s, mean_only):
        assert np.array_equal(np.isneginf(got), ~fin)
        np.testing.assert_allclose(got[fin], mean_p[fin], rtol=1e-10, atol=1e-10)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    assert np.max(np.abs(std_s ** 2 - std_p ** 2)) <= 1e-11 * C
    assert np.max(np.abs(std_s[mask != _lib.MASK_CLASSIFIED_INF] ** 2 - rs[mask != _lib.MASK_CLASSIFIED_INF] ** 2)) <= 1e-9 * C
    if M > 2:
        assert std_s[1] == 0.0 and std_s[2] > 0.0


def test_capacity_does_not_grow_when_only_the_dimension_changes(dev):
    """Regression (found by tests/tools/fuzz_parity.py): alternating small / large d used to enlarge the
    N x N buffers by 12.5 % each time until the device ran out of memory."""
    rng = np.random.default_rng(0)
    for it in range(80):
        d = 32 if it % 2 else 3
        X = rng.uniform(0, 1, (200, d))
        dev.set_train(X, rng.standard_normal(200), np.full(200, 1e-3))
        dev.set_theta(3, np.log(np.array([2.0] + [0.5] * d)))
    dev.set_affine(None, None, 0.0, 1.0, np.inf)          # (the fixture carries the map of the test before)
    assert dev.factorize() == 0
    m = dev.predict(rng.uniform(0, 1, (5, d)))
    assert np.all(np.isfinite(m))
    with pytest.raises(ValueError):
        dev.predict(rng.uniform(0, 1, (5, 3)))          # the model holds d = 32 columns


def test_f5_logexp_edge_vectors_through_the_device_epilogue(dev):
    """The reference's own F5 vectors (incl. sigma <= sigma_n and mu = -inf rows) fed straight to the
    acquisition epilogue of the sweep (gpry/acquisition_functions.py:1068-1074)."""
    g = load_golden("predict")
    zeta, base, noise = float(g["f5_zeta"]), float(g["f5_baseline"]), float(g["f5_noise"])
    acq = dev.debug_logexp(g["f5_mu"], g["f5_std"], zeta, base, noise)
    ref = g["f5_acq"]
    assert np.isneginf(ref).sum() >= 2                       # the fixture does hold edge rows
    assert np.array_equal(np.isneginf(acq), np.isneginf(ref)) and not np.isnan(acq).any()
    fin = np.isfinite(ref)
    np.testing.assert_allclose(acq[fin], ref[fin], rtol=1e-14, atol=1e-14)
    # a few more edges: sigma exactly sigma_n, sigma = 0, mu = -inf with a large sigma, +inf baseline gap
    mu = np.array([1.0, 2.0, -np.inf, 0.5])
    sd = np.array([noise, 0.0, 3.0, 1e-300])
    got = dev.debug_logexp(mu, sd, zeta, base, noise)
    assert np.all(np.isneginf(got))


@pytest.mark.timeout(600)
def test_full_size_config2_sweep_topk_and_multi_add_vs_oracle():
    """BASELINE configs[2] at full size: N=4096, d=16, Matern-5/2, M=1e6 resident pool, default chunking
    (31 chunks).  (a) the device shortlist is the head of np.lexsort on the fetched acquisition and
    `bound` is the first value outside it; (b) mean / std / acquisition of a random 2048-row subset
    against the oracle; (c) NORA.multi_add(n_points=16) equals the oracle's restatement of the
    reference (gpry/gp_acquisition.py:971-1108) run on the device shortlist united with 50 000 random
    rows of the pool -- every candidate that can enter the ranked pool is in that union."""
    from gpry_amd.gp_acquisition import NORA
    from test_host_mirror_gpu import make_gpr
    N, d, M, npts = 4096, 16, 1_000_000, 16
    bounds, X, y, Xc = orc.synthetic_problem(N, d, M)
    theta = np.log(np.array([4.0] + [0.3] * d))
    gpr = make_gpr(bounds, 3, theta=theta)
    gpr.append_to_data(X, y, fit_gpr=False)
    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    dev = gpr.device
    out = dev.sweep_fetch(("y", "sigma", "acq"))
    # (a) exact selection over 1e6 candidates
    a = out["acq"]
    assert len(a) == M and not np.isnan(a).any()
    order = np.lexsort((-np.arange(M), -a))
    top, bound = dev.sweep_topk(256)
    np.testing.assert_array_equal(top["idx"], order[:256])
    np.testing.assert_array_equal(top["acq"], a[order[:256]])
    np.testing.assert_array_equal(top["y"], out["y"][order[:256]])
    np.testing.assert_array_equal(top["sigma"], out["sigma"][order[:256]])
    assert bound == a[order[256]]
    # (b) a random subset against the oracle (same tolerances as the small-size tests)
    ref = orc.OracleGPR(bounds, kernel_id=orc.MATERN52)
    ref.theta = theta
    ref.fitted = True
    ref.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(M, 2048, replace=False))
    rm, rs = ref.predict(Xc[sub], return_std=True)
    C = np.exp(theta[0]) * ref.pre_y.std_ ** 2
    assert np.max(np.abs(out["y"][sub] - rm)) <= 1e-8 * np.max(np.abs(rm))
    assert np.max(np.abs(out["sigma"][sub] ** 2 - rs ** 2)) <= 1e-9 * C
    zeta = orc.auto_zeta(d)
    racq = orc.logexp_f(rm, rs, ref.y_max, ref.noise_level, zeta)
    assert np.array_equal(np.isneginf(a[sub]), np.isneginf(racq))
    fin = np.isfinite(racq)
    np.testing.assert_allclose(a[sub][fin], racq[fin], rtol=1e-6, atol=1e-6)
    # (c) the proposals
    union = np.union1d(order[:max(256, acq.stats["shortlist"])], rng.choice(M, 50_000, replace=False))
    Xr, yr, ar = orc.nora_multi_add(ref, Xc[union], npts)
    np.testing.assert_array_equal(Xp, Xr)
    np.testing.assert_allclose(yp, yr, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(ap, ar, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,d,M", [(1000, 7, 17), (1000, 7, 130), (4096, 16, 64), (4096, 16, 1000), (2500, 5, 3000)])
def test_split_k_contraction_of_small_batches_equals_the_one_pass_contraction(dev, N, d, M):
    """predict(return_std=True) for 17 ... a few thousand points splits every tile's k-range over
    several workgroups (partial products in scratch, squared after a fixed-order sum): same
    variances as the one-pass contraction (SUMSQ epilogue) to rounding, and as the oracle."""
    from gpry_amd import _lib
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=29)
    m = orc.OracleGPR(bounds, kernel_id=3)
    m.theta = np.log(np.array([4.0] + [0.3] * d))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    mask = np.zeros(M, dtype=np.uint8)
    mask[1], mask[2] = _lib.MASK_CLASSIFIED_INF, _lib.MASK_OUTSIDE_TRUST
    try:
        dev.set_option("predict_split", 0)
        mean_1, std_1 = dev.predict(Xc, return_std=True, mask=mask)
        dev.timing_reset()
        dev.set_option("predict_split", 1)
        mean_s, std_s = dev.predict(Xc, return_std=True, mask=mask)
        assert dev.timing("sweep_gemm_splitk")[1] == 1 and dev.timing("sweep_gemm")[1] == 0   # the path ran
    finally:
        dev.set_option("predict_split", 1)
        dev.set_option("timing", 0)
    np.testing.assert_array_equal(mean_s, mean_1)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    assert np.max(np.abs(std_s ** 2 - std_1 ** 2)) <= 1e-12 * C
    assert std_s[1] == 0.0 and std_s[2] > 0.0 and np.isneginf(mean_s[1]) and np.isneginf(mean_s[2])
    rm, rs = m.predict(Xc, return_std=True)
    keep = mask != _lib.MASK_CLASSIFIED_INF
    assert np.max(np.abs(std_s[keep] ** 2 - rs[keep] ** 2)) <= 1e-9 * C


@pytest.mark.parametrize("N", [2048, 2300, 4096])
def test_sweep_with_alternating_k_walk_is_chunking_independent(dev, N):
    """The super-tiles of an XCD alternate the direction of their k walk so that the row tiles of a super-tile sit at
    the same k and share the K*^T panel in L2 (profiles/HISTORY.md section 4.1(b)).  The direction is a function of the row tile
    alone: the bits of a candidate's sigma must not depend on the chunking (which moves its column inside a launch and
    switches between the paired and the plain super-tile order), the mean is untouched, and against the register-staged
    engine (``gemm_dma`` = 0: every tile walks k upwards) the variance moves by rounding only; both agree with the oracle."""
    d, M = 6, 21000
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=41)
    m = orc.OracleGPR(bounds, kernel_id=3)
    m.theta = np.log(np.array([4.0] + [0.3] * d))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    res = {}
    try:
        for dma in (1, 0):
            for chunk in (8192, 5120, 1024):
                dev.set_option("gemm_dma", dma)
                dev.set_option("sweep_chunk", chunk)
                out = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("y", "sigma"))
                res[dma, chunk] = (out["y"].copy(), out["sigma"].copy())
    finally:
        dev.set_option("gemm_dma", 1)
        dev.set_option("sweep_chunk", 0)
    for dma in (1, 0):
        for chunk in (5120, 1024):
            np.testing.assert_array_equal(res[dma, chunk][1], res[dma, 8192][1])
            np.testing.assert_array_equal(res[dma, chunk][0], res[dma, 8192][0])
    np.testing.assert_array_equal(res[1, 8192][0], res[0, 8192][0])
    assert np.max(np.abs(res[1, 8192][1] ** 2 - res[0, 8192][1] ** 2)) <= 1e-13 * C
    rm, rs = m.predict(Xc[:3000], return_std=True)
    assert np.max(np.abs(res[1, 8192][1][:3000] ** 2 - rs ** 2)) <= 1e-9 * C


def test_f8_bordered_append_vs_reference(dev):
    """F8 through gpry_append_rows: the factor of the first 32 points is extended by border rows for the
    three appended ones (fixed theta, frozen pre-processors) and must match what the reference gets by
    rebuilding K and refactorising (gpry/gpr.py:1015-1017)."""
    g = load_golden("predict")
    m = _oracle_model(g, "f8_", 2)
    X, y, Xc = g["f8_X"], g["f8_y"], g["f8_Xc"]
    m.append_to_data(X[:32], y[:32], fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)                                     # 32-point model, factorised
    m.append_to_data(X[32:], y[32:], fit_gpr=False, fit_preprocessors=False)
    assert np.array_equal(m.X_train_[:32], dev_rows := m.X_train_[:32]) and len(m.y_train_) == 35
    assert dev.append_rows(m.X_train_[32:], m.y_train_[32:], m.alpha[32:]) == 0
    assert dev.N == 35
    dev.set_affine(m.pre_X.lo, m.pre_X.hi - m.pre_X.lo, m.pre_y.mean_, m.pre_y.std_, m.clip_hi())
    L, V, a = dev.get_factor()
    np.testing.assert_allclose(L, g["f8_L"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(V, g["f8_V"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(a, g["f8_alpha_"], rtol=1e-8, atol=1e-9)
    assert np.all(np.triu(L, 1) == 0.0) and np.all(np.triu(V, 1) == 0.0)
    mean, std = dev.predict(Xc, return_std=True)
    np.testing.assert_allclose(mean, g["f8_mean_after"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(std, g["f8_std_after"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("N,k,d,kid", [(100, 1, 3, 0), (120, 20, 4, 3), (128, 1, 2, 1), (250, 70, 5, 2),
                                       (1000, 8, 8, 3), (4090, 16, 16, 3)])
def test_bordered_append_equals_refactorisation(dev, N, k, d, kid):
    """gpry_append_rows against gpry_factorize on the enlarged set: sizes that stay inside the padded
    size, cross it (120 + 20, 128 + 1, 4090 + 16), need more than one 64-row chunk (70) or new buffers;
    then an LML evaluation and a second append on top of the first."""
    rng = np.random.default_rng(N + k)
    X = rng.uniform(0, 1, (N + k + 2, d))
    y = np.sin(3 * X).sum(1) + 0.05 * rng.standard_normal(len(X))
    alpha = np.full(len(X), 1e-4)
    theta = np.log(np.array([2.0] + [0.35] * d))
    dev.set_affine()                                  # the shared context may carry another test's maps
    dev.set_train(X[:N + k], y[:N + k], alpha[:N + k])
    dev.set_theta(kid, theta)
    assert dev.factorize() == 0
    L1, V1, a1 = dev.get_factor()
    lml1 = dev.lml(theta, True)
    dev.set_train(X[:N], y[:N], alpha[:N])
    dev.set_theta(kid, theta)
    assert dev.factorize() == 0
    assert dev.append_rows(X[N:N + k], y[N:N + k], alpha[N:N + k]) == 0
    L2, V2, a2 = dev.get_factor()
    assert L2.shape == (N + k, N + k)
    scale = np.max(np.abs(L1))
    assert np.max(np.abs(L2 - L1)) <= 1e-11 * scale
    assert np.max(np.abs(V2 - V1)) <= 1e-9 * np.max(np.abs(V1))
    assert np.max(np.abs(a2 - a1)) <= 1e-8 * np.max(np.abs(a1))
    assert np.all(np.triu(L2, 1) == 0.0) and np.all(np.triu(V2, 1) == 0.0)
    assert np.max(np.abs(V2 @ L2 - np.eye(N + k))) < 1e-8
    Xc = rng.uniform(0, 1, (40, d))
    m2, s2 = dev.predict(Xc, return_std=True)
    lml2 = dev.lml(theta, True)                       # works on the enlarged training set
    assert abs(lml2[0] - lml1[0]) <= 1e-10 * abs(lml1[0])
    np.testing.assert_allclose(lml2[1], lml1[1], rtol=1e-7, atol=1e-7 * np.max(np.abs(lml1[1])))
    # a second append on top of the bordered factor, against a fresh factorisation of everything
    assert dev.append_rows(X[N + k:], y[N + k:], alpha[N + k:]) == 0
    L3, V3, a3 = dev.get_factor()
    m3, s3 = dev.predict(Xc, return_std=True)
    dev.set_train(X, y, alpha)
    dev.set_theta(kid, theta)
    assert dev.factorize() == 0
    L4, V4, a4 = dev.get_factor()
    m4, s4 = dev.predict(Xc, return_std=True)
    assert np.max(np.abs(L3 - L4)) <= 1e-11 * scale and np.max(np.abs(a3 - a4)) <= 1e-8 * np.max(np.abs(a4))
    np.testing.assert_allclose(m3, m4, rtol=1e-9, atol=1e-9)
    assert np.max(np.abs(s3 ** 2 - s4 ** 2)) <= 1e-10 * np.exp(theta[0])


def test_bordered_append_reports_a_non_positive_definite_border(dev):
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 1, (50, 2))
    y = rng.standard_normal(50)
    dev.set_affine()
    dev.set_train(X, y, np.full(50, 1e-10))
    dev.set_theta(0, np.log(np.array([1.0, 5.0, 5.0])))
    if dev.factorize() != 0:
        pytest.skip("base matrix already singular")
    # an exact duplicate of a training point whose diagonal entry is pushed down: the Schur complement
    # C - u^T u of the border is negative
    info = dev.append_rows(X[:1], y[:1], np.array([-0.5]))
    assert info == 51                                  # 1-based column of the failing pivot
    with pytest.raises(Exception):
        dev.predict(X[:3], return_std=True)           # no valid factor any more: the caller must refactorise


@pytest.mark.parametrize("N", [100, 200, 300, 1100, 2100, 3100, 4096, 5000, 6100, 7300, 8192])
def test_cholesky_with_tiles_riding_in_the_panel_launches_is_bit_identical(dev, N):
    """Default schedule: the trailing update is cut into 64 x 64 tiles that ride as extra workgroups in the panel
    launches (deadline-driven plan), the panel steps apply the previous strip to their own columns themselves.
    Above Np = 3584 the matrix is a list of segments: outer blocks of up to 768 columns, each with the riding tiles of
    its own columns and one SYRK launch behind it for everything to the right, then the last 3584 columns.  Same updates, same order, same arithmetic as the schedule with
    separate trailing launches: the factor must be bit-identical; a non-positive-definite matrix must report
    the same leading minor (in the tail: as a column of the whole matrix)."""
    d = 4
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d))
    y = rng.standard_normal(N)
    dev.set_affine()
    dev.set_train(X, y, np.full(N, 1e-5))
    theta = np.log(np.array([3.0, 0.4, 0.5, 0.6, 0.7]))
    dev.set_theta(3, theta)
    try:
        # (the schedule that also delivers the inverse factor -- "chol_stacked", next test -- has no separate-launch twin)
        dev.set_option("chol_stacked", 0)
        dev.set_option("chol_overlap", 0)           # every trailing update its own launch
        assert dev.factorize() == 0
        L0, V0, a0 = dev.get_factor()
        lml0 = dev.lml(theta, True)
        dev.set_option("chol_overlap", 1)
        for _ in range(2):
            assert dev.factorize() == 0
            L1, V1, a1 = dev.get_factor()
            assert np.array_equal(L0, L1) and np.array_equal(V0, V1) and np.array_equal(a0, a1)
        lml1 = dev.lml(theta, True)
        assert lml0[0] == lml1[0] and np.array_equal(lml0[1], lml1[1])
        # Round 6: every element of the factor is ONE chain of MFMAs over the columns left of it, started at the covariance
        # entry, whatever launch its pieces ride in -- so the column blocks of the throughput schedule (any widths, one SYRK
        # launch behind each; "chol_tp_segments") give this very factor too, bit for bit, and the same failing minor below
        # ... left-looking (one deep launch in front of every block: "tp_left" = 1) as well as right-looking (one launch of the
        # block's width behind every block)
        for blk, tail, left in ((512, 1024, 1), (512, 1024, 0), (128, 128, 1), (256, 640, 0), (256, 640, 1)):
            dev.set_option("tp_block", blk); dev.set_option("tp_tail", tail); dev.set_option("tp_left", left)
            dev.set_option("chol_tp_segments", 1)
            assert dev.factorize() == 0
            L2, V2, a2 = dev.get_factor()
            assert np.array_equal(L0, L2) and np.array_equal(V0, V2) and np.array_equal(a0, a2), (blk, tail, left)
        dev.set_option("chol_tp_segments", 0)
        K = dev.kernel_train(add_alpha=True)
        assert relmax(L1 @ L1.T, K) < 1e-13
        # not positive definite: a duplicated row with zero noise far down the matrix
        if N >= 300:
            Xb = X.copy()
            Xb[N - 7] = Xb[N // 3]
            alpha = np.full(N, 1e-5)
            alpha[N - 7] = alpha[N // 3] = -1e-3
            infos = []
            for ov in (0, 1, 2):
                dev.set_option("chol_overlap", min(ov, 1))
                dev.set_option("chol_tp_segments", 1 if ov == 2 else 0)
                dev.set_train(Xb, y, alpha)
                dev.set_theta(3, theta)
                infos.append(dev.factorize())
            dev.set_option("chol_tp_segments", 0)
            assert infos[0] == infos[1] == infos[2] and infos[0] > 0
        # ... and inside an outer block of the large schedule (its column counts from the start of the whole matrix)
        if N >= 5000:
            Xb = X.copy()
            Xb[700] = Xb[300]
            alpha = np.full(N, 1e-5)
            alpha[700] = alpha[300] = -1e-3
            infos = []
            for ov in (0, 1):
                dev.set_option("chol_overlap", ov)
                dev.set_train(Xb, y, alpha)
                dev.set_theta(3, theta)
                infos.append(dev.factorize())
            assert infos[0] == infos[1] and 300 < infos[0] <= 701
    finally:
        dev.set_option("chol_overlap", 1)
        dev.set_option("chol_stacked", 2048)
        dev.set_option("chol_tp_segments", 0)
        dev.set_option("tp_block", 512); dev.set_option("tp_tail", 1024); dev.set_option("tp_left", 0)


def test_a_panel_step_that_timed_out_is_an_error_not_a_verdict_on_the_matrix(dev):
    """the bounded waits of the panel step report a timeout through the status word of a not-positive-definite
    matrix plus a marker in the fourth word.  The host reads the marker: the caller gets an error (-2, "timed out"), never the
    -inf of sklearn's non-PD convention (sklearn:_gpr.py:586-589), which would let an optimiser go on with a wrong
    objective.  The marker is forced with the test hook "panel_debug" = 64."""
    from gpry_amd import _lib
    rng = np.random.default_rng(3)
    N, d = 300, 3
    X = rng.uniform(size=(N, d)); y = rng.standard_normal(N)
    dev.set_affine()
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([2.0, 0.5, 0.5, 0.5]))
    dev.set_theta(3, theta)
    ok = dev.lml(theta, True)
    assert np.isfinite(ok[0])
    try:
        dev.set_option("panel_debug", 64)
        with pytest.raises(_lib.GpryHipError, match="timed out"):
            dev.lml(theta, True)
        with pytest.raises(_lib.GpryHipError, match="timed out"):
            dev.lml_batch(np.array([theta, theta + 0.1]), True)
        with pytest.raises(_lib.GpryHipError, match="timed out"):
            dev.factorize()
        for sched in (1,):
            dev.set_option("lml_schedule", sched)
            with pytest.raises(_lib.GpryHipError, match="timed out"):
                dev.lml_batch(np.array([theta, theta + 0.1, theta - 0.1]), True)
    finally:
        dev.set_option("panel_debug", 0)
        dev.set_option("lml_schedule", 0)
    again = dev.lml(theta, True)
    assert again[0] == ok[0] and np.array_equal(again[1], ok[1])


@pytest.mark.parametrize("N", [130, 200, 500, 1000, 1100, 2048, 2100, 3100, 3584])
def test_inverse_factor_from_the_rows_appended_to_the_cholesky(dev, N):
    """Up to Np = "chol_stacked" (default 2048; here also forced above it, up to the largest single-segment size) the inverse
    factor V = L^-1 (gpry/gpr.py:1456-1457: solve_triangular(L, I)) is not computed behind the Cholesky: the identity is appended
    to the matrix as extra rows, [K; I], and the panel chain -- same step, same riding tiles -- leaves L^-T in their place
    (potrf_stacked, csrc/chol_panel.hip), which is transposed into V.  L must be the factor of the other schedule bit for
    bit; V the inverse of L to rounding (a different summation order than the recursive inverse), lower triangular with exact
    zeros above the diagonal; alpha_, the LML and its gradient those of the recursive path to rounding and the oracle's within
    the tolerances of the objective tests; repeated calls and the adoption of an objective evaluation's factor give the same
    bits; a matrix that is not positive definite reports the same leading minor."""
    d = 5
    rng = np.random.default_rng(N + 3)
    X = rng.uniform(0, 1, (N, d))
    y = np.sin(3 * X).sum(1) + 0.05 * rng.standard_normal(N)
    noise = np.full(N, 1e-5)
    dev.set_affine()
    dev.set_train(X, y, noise)
    theta = np.log(np.array([3.0, 0.4, 0.5, 0.6, 0.7, 0.8]))
    dev.set_theta(3, theta)
    try:
        dev.set_option("lml_cache", 0)
        dev.set_option("chol_stacked", 0)
        assert dev.factorize() == 0
        L0, V0, a0 = dev.get_factor()
        lml0 = dev.lml(theta, True)
        dev.set_option("chol_stacked", 4096)
        for _ in range(2):
            assert dev.factorize() == 0
            L1, V1, a1 = dev.get_factor()
            if _ == 0:
                first = (V1, a1)
            assert np.array_equal(V1, first[0]) and np.array_equal(a1, first[1])
        assert np.array_equal(L0, L1)
        assert not np.triu(V1, 1).any()
        Ld = np.tril(L1)
        assert relmax(np.tril(V1) @ Ld, np.eye(N)) < 1e-9
        assert relmax(V1, V0) < 1e-11 and relmax(a1, a0) < 1e-9
        lml1 = dev.lml(theta, True)
        assert abs(lml1[0] - lml0[0]) <= 1e-12 * max(1.0, abs(lml0[0]))
        assert np.max(np.abs(lml1[1] - lml0[1])) <= 1e-9 * max(1.0, np.max(np.abs(lml0[1])))
        if N <= 2100:
            rl, rg = orc.log_marginal_likelihood(X, y, noise, theta, 3, eval_gradient=True)
            assert abs(lml1[0] - rl) <= 1e-10 * max(1.0, abs(rl))
            assert np.max(np.abs(lml1[1] - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg)))
        # the comparator: the appended rows without use of their zeros (every row block in every step, every panel on every
        # tile) -- skipping exact zeros changes no bit
        if N <= 2100:
            dev.set_option("chol_stacked_dense", 1)
            assert dev.factorize() == 0
            L3, V3, a3 = dev.get_factor()
            dev.set_option("chol_stacked_dense", 0)
            assert np.array_equal(L3, L1) and np.array_equal(V3, V1) and np.array_equal(a3, a1)
        # the factor of an objective evaluation, adopted by factorize: the same bits
        dev.set_option("lml_cache", 1)
        dev.lml(theta, True)
        assert dev.factorize() == 0
        L2, V2, a2 = dev.get_factor()
        assert np.array_equal(L2, L1) and np.array_equal(V2, V1) and np.array_equal(a2, a1)
        # a batch: every theta the bits of its single evaluation
        thetas = theta + rng.uniform(-0.3, 0.3, (3, d + 1))
        single = [dev.lml(th, True) for th in thetas]
        lb, gb, ib = dev.lml_batch(thetas, True)
        for b, (l1, g1, i1) in enumerate(single):
            assert lb[b] == l1 and ib[b] == i1 == 0
            np.testing.assert_array_equal(gb[b], g1)
        # not positive definite: a duplicated row with negative noise
        if N >= 300:
            Xb = X.copy(); Xb[N - 7] = Xb[N // 3]
            nb = noise.copy(); nb[N - 7] = nb[N // 3] = -1e-3
            infos = []
            for st in (0, 4096):
                dev.set_option("chol_stacked", st)
                dev.set_train(Xb, y, nb)
                dev.set_theta(3, theta)
                infos.append(dev.factorize())
            assert infos[0] == infos[1] and infos[0] > 0
    finally:
        dev.set_option("chol_stacked", 2048)
        dev.set_option("chol_stacked_dense", 0)
        dev.set_option("lml_cache", 1)


@pytest.mark.parametrize("N", [1000, 1100, 1300, 1700, 2100, 4096, 5000, 6100, 7300])
def test_pipelined_factor_chain_is_bit_identical(dev, N):
    """V = L^-1 queued phase by phase on a second stream underneath the Cholesky panel chain
    (``factor_pipeline=1``, default from Np = 1280): same products, same split-K factors, same operands as
    the serial chain -- L, V, alpha_, the LML and its gradient must be bit-identical, on power-of-two and
    ragged block counts, with both Cholesky schedules, and a non-positive-definite matrix must report the
    same leading minor."""
    d = 5
    rng = np.random.default_rng(N + 1)
    X = rng.uniform(0, 1, (N, d))
    y = rng.standard_normal(N)
    dev.set_affine()
    dev.set_train(X, y, np.full(N, 1e-5))
    theta = np.log(np.array([3.0, 0.4, 0.5, 0.6, 0.7, 0.8]))
    dev.set_theta(3, theta)
    try:
        dev.set_option("factor_pipeline_min", 0)        # default: from Np = 1280 on
        dev.set_option("chol_stacked", 0)               # (where V comes out of the Cholesky launches there is nothing to pipeline)
        for overlap in (1, 0):
            dev.set_option("chol_overlap", overlap)
            dev.set_option("factor_pipeline", 0)
            assert dev.factorize() == 0
            L0, V0, a0 = dev.get_factor()
            lml0 = dev.lml(theta, True)
            dev.set_option("factor_pipeline", 1)
            for _ in range(3):
                assert dev.factorize() == 0
                L1, V1, a1 = dev.get_factor()
                assert np.array_equal(L0, L1) and np.array_equal(V0, V1) and np.array_equal(a0, a1)
                lml1 = dev.lml(theta, True)
                assert lml0[0] == lml1[0] and np.array_equal(lml0[1], lml1[1])
        Vd = np.tril(V1)[:N, :N]
        Ld = np.tril(L1)[:N, :N]
        assert relmax(Vd @ Ld, np.eye(N)) < 1e-9
        Xb = X.copy()
        Xb[N - 7] = Xb[N // 3]
        alpha = np.full(N, 1e-5)
        alpha[N - 7] = alpha[N // 3] = -1e-3
        infos = []
        for pipe in (0, 1):
            dev.set_option("factor_pipeline", pipe)
            dev.set_train(Xb, y, alpha)
            dev.set_theta(3, theta)
            infos.append(dev.factorize())
        assert infos[0] == infos[1] and infos[0] > 0
    finally:
        dev.set_option("chol_overlap", 1)
        dev.set_option("factor_pipeline", 1)
        dev.set_option("factor_pipeline_min", 1280)
        dev.set_option("chol_stacked", 2048)



def test_options_from_the_environment(monkeypatch):
    """``GPRY_HIP_OPTIONS="key=value,..."`` reaches every new context (A/B runs of unmodified callers); a
    malformed entry or an unknown key fails the creation loudly instead of being ignored."""
    from gpry_amd import _lib
    monkeypatch.setenv("GPRY_HIP_OPTIONS", "factor_pipeline=0,chol_overlap=1")
    d = _lib.Device(0)
    d.close()
    for bad in ("factor_pipeline", "no_such_option=1", "chol_overlap=x", "=3"):
        monkeypatch.setenv("GPRY_HIP_OPTIONS", bad)
        with pytest.raises(_lib.GpryHipError, match="GPRY_HIP_OPTIONS"):
            _lib.Device(0)
    monkeypatch.delenv("GPRY_HIP_OPTIONS")
    _lib.Device(0).close()


@pytest.mark.parametrize("seg", [1, 3, 7, 20, 1000])
def test_stream_k_gemm_matches_numpy(dev, seg):
    """Stream-K launches of the DMA engine (V = L^-1 levels and K^-1 = V^T V): the (tile, k) space cut into
    segments of `seg` slab pairs -- parts that start and end inside tiles, whole tiles, tiles in one part -- in the
    three layouts and triangular modes the factor chain uses, both store epilogues."""
    rng = np.random.default_rng(40 + seg)
    n = 640
    L = np.tril(rng.standard_normal((n, n)))
    G = rng.standard_normal((n, n))
    hook = seg << 16
    # T = L21 V11: B lower (k >= tj * 128);  V21 = -V22 T: A lower (k < (ti + 1) * 128), negated store
    assert relmax(dev.debug_gemm(G, L, None, n, n, n, kmode=2, tile_map=hook), G @ L) < 1e-14
    assert relmax(dev.debug_gemm(L, G, None, n, n, n, kmode=1, epi=1, tile_map=hook), -(L @ G)) < 1e-14
    # K^-1 = V^T V, lower tiles only
    C0 = rng.standard_normal((n, n))
    got = dev.debug_gemm(L, L, C0, n, n, n, a_trans=1, kmode=3, lower_only=True, tile_map=hook)
    assert relmax(np.tril(got), np.tril(L.T @ L)) < 1e-14
    np.testing.assert_array_equal(got[:128, 128:], C0[:128, 128:])        # upper tiles are not touched
    # rectangular, full k-range, NT layout
    A = rng.standard_normal((256, 384))
    B = rng.standard_normal((384, 512))
    assert relmax(dev.debug_gemm(A, np.ascontiguousarray(B.T), None, 256, 512, 384, b_trans=1, tile_map=hook), A @ B) < 1e-14



def test_kb_session_after_the_padded_size_grew_within_the_allocation():
    """ADVICE r02 (append.hip / ctx.hip): the Kriging-believer row buffer was sized kb_cap x Np at its
    allocation; when the training set then grew to a wider padded size WITHOUT new buffers (Np 1280 ->
    1408 inside cap = 1408), a session of ~1000 rows wrote beyond its end.  The conditioned variances of
    all registered points are compared with predict() here, after exactly that sequence."""
    from gpry_amd import _lib
    rng = np.random.default_rng(77)
    d = 4
    X = rng.uniform(size=(1400, d))
    y = np.sin(X.sum(axis=1))
    theta = np.log(np.array([2.0] + [0.4] * d))
    dv = _lib.Device(0)
    dv_round = 0
    try:
        for n in (1400, 1200, 1400):            # cap = 1408; Np = 1408 -> 1280 -> 1408 in the same buffers
            dv.set_train(X[:n], y[:n], 1e-4)
            dv.set_theta(3, theta)
            assert dv.factorize() == 0
            dv.set_affine()
            if dv_round == 0:                   # only sizes the N x N buffers (cap = 1408)
                dv_round += 1
                continue
            m = 1000 if n == 1400 else 64       # the small session sizes the row buffer for Np = 1280
            Xc = rng.uniform(size=(m, d))
            first, var0 = dv.kb_register(Xc)
            assert first == 0
            _, sd = dv.predict(Xc, return_std=True)
            assert np.max(np.abs(var0 - sd ** 2)) <= 1e-9 * np.exp(theta[0])
            G, kv = dv.kb_gram(m - 1, m)        # reads the last row: beyond the old allocation before the fix
            assert np.isfinite(G).all() and abs(G[m - 1] - (np.exp(theta[0]) - var0[m - 1])) <= 1e-9 * np.exp(theta[0])
    finally:
        dv.close()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("N,d,form", [(4096, 16, "einsum"), (8192, 20, "blocked")])
def test_full_size_objective_and_gradient_vs_oracle(dev, N, d, form):
    """the hyper-parameter objective of BASELINE configs[2] / configs[4] against the ORACLE at full
    size (until now the device LML above N = 1500 was only compared with itself through finite differences and
    L L^T = K).  N=4096, d=16: value and gradient against the einsum restatement of sklearn:_gpr.py:574-652
    (~10 GB of (N, N, 1+d) tensors on the host).  N=8192, d=20: value and gradient against the blocked,
    memory-light restatement (``orc.log_marginal_likelihood_blocked``, itself checked against the reference's F3
    vectors and the einsum form in tests/test_oracle_golden.py).  Tolerances as in the small tests: LML rel
    1e-10, gradient 1e-7 of its largest entry.  Two thetas: the bench's (C=4, l=0.3) and an anisotropic one."""
    bounds, X, y, _ = orc.synthetic_problem(N, d, 8)
    pre = orc.NormalizeBounds(bounds)
    X_ = pre.transform(X)
    ym, ys = y.mean(), y.std()
    y_ = (y - ym) / ys
    alpha = np.full(N, (1e-2 / ys) ** 2)
    dev.set_train(X_, y_, alpha)
    rng = np.random.default_rng(N)
    thetas = [np.log(np.array([4.0] + [0.3] * d)),
              np.log(np.concatenate(([9.0], 0.25 * (1.0 + rng.uniform(size=d)))))]
    for it, theta in enumerate(thetas):
        dev.set_theta(3, theta)
        lml, grad, info = dev.lml(theta, True)
        assert info == 0
        if form == "einsum" and it == 0:
            rl, rg = orc.log_marginal_likelihood(X_, y_, alpha, theta, orc.MATERN52, eval_gradient=True)
        else:
            rl, rg = orc.log_marginal_likelihood_blocked(X_, y_, alpha, theta, orc.MATERN52)
        assert abs(lml - rl) <= 1e-10 * abs(rl), (lml, rl)
        assert np.max(np.abs(grad - rg)) <= 1e-7 * np.max(np.abs(rg)), (grad, rg)
        assert dev.lml(theta, False)[0] == lml          # the value-only evaluation gives the same bits


@pytest.mark.timeout(900)
def test_full_size_config1_multi_add_vs_oracle():
    """BASELINE configs[1] (8-d correlated Gaussian, N=1024, anisotropic RBF) with a pool of M = 1e5 candidates:
    ``NORA.multi_add(n_points=8)`` equals the oracle's restatement of the reference (gpry/gp_acquisition.py:971-1108)
    run over ALL 1e5 candidates -- proposals identical, lies and acquisition values within the small-test
    tolerances -- and y / sigma of every candidate within 1e-8 / 1e-9 C."""
    from gpry_amd.gp_acquisition import NORA
    from test_host_mirror_gpu import make_gpr
    N, d, M, npts = 1024, 8, 100_000, 8
    bounds, X, y, Xc = orc.synthetic_problem(N, d, M)
    theta = np.log(np.array([4.0, 0.3, 0.25, 0.4, 0.35, 0.3, 0.5, 0.28, 0.33]))
    gpr = make_gpr(bounds, 0, theta=theta)
    gpr.append_to_data(X, y, fit_gpr=False)
    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    ref = orc.OracleGPR(bounds, kernel_id=orc.RBF)
    ref.theta = theta
    ref.fitted = True
    ref.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    Xr, yr, ar, allr = orc.nora_multi_add(ref, Xc, npts, return_all=True)
    np.testing.assert_array_equal(Xp, Xr)
    np.testing.assert_allclose(yp, yr, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(ap, ar, rtol=1e-5, atol=1e-6)
    out = gpr.device.sweep_fetch(("y", "sigma", "acq"))
    C = np.exp(theta[0]) * ref.pre_y.std_ ** 2
    assert np.max(np.abs(out["y"] - allr["y"])) <= 1e-8 * np.max(np.abs(allr["y"]))
    assert np.max(np.abs(out["sigma"] ** 2 - allr["sigma"] ** 2)) <= 1e-9 * C
    assert int(np.argmax(out["acq"])) == int(np.argmax(allr["acq"]))          # acquisition arg-max identical


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("N,d,ls", [(700, 3, 0.3), (1500, 8, 0.05), (3000, 16, 0.3), (2100, 20, 1.5), (900, 5, 0.01)])
def test_cross_kernel_panel_with_distances_from_the_matrix_pipe(dev, N, d, ls, kid):
    """Round 4: the sweep's K*^T panel (gpry/gpr.py:1179) takes r^2 = |x - c|^2 + |y - c|^2 - 2 (x - c).(y - c) with the dot
    products on the matrix pipe (``cross_build_mfma_kernel``, option ``cross_mfma``; c = mean of the training rows, so that
    the cancellation is relative to the spread of the data).  Against the difference form (``cross_mfma`` = 0) and against
    the oracle: mean rel <= 1e-8, |delta var| <= 1e-9 C -- also with length scales of 1 % of the box, with a candidate ON a
    training point (r^2 = 0 comes out as rounding noise and is clamped) and a ragged chunk.  Matern-1/2 keeps the difference
    form whatever the option says (the cusp of exp(-r) at r = 0 turns noise e in r^2 into sqrt(e) in k): there the two
    runs must agree bit for bit."""
    M = 5000 + 13
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=N + kid)
    Xc[7] = X[11]                                   # a candidate on a training point
    m = orc.OracleGPR(bounds, kernel_id=kid)
    m.theta = np.log(np.array([3.0] + [ls] * d))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    C = np.exp(m.theta[0]) * m.pre_y.std_ ** 2
    out = {}
    try:
        for mf in (1, 0):
            dev.set_option("cross_mfma", mf)
            dev.set_option("sweep_chunk", 2048)      # several chunks, the last one ragged
            r = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("y", "sigma"))
            out[mf] = (r["y"].copy(), r["sigma"].copy())
    finally:
        dev.set_option("cross_mfma", 1)
        dev.set_option("sweep_chunk", 0)
    scale = max(1.0, np.max(np.abs(out[0][0])))
    if kid == 1:
        np.testing.assert_array_equal(out[1][0], out[0][0])
        np.testing.assert_array_equal(out[1][1], out[0][1])
    assert np.max(np.abs(out[1][0] - out[0][0])) <= 1e-9 * scale
    assert np.max(np.abs(out[1][1] ** 2 - out[0][1] ** 2)) <= 1e-10 * C
    rm, rs = m.predict(Xc[:1500], return_std=True)
    assert np.max(np.abs(out[1][0][:1500] - rm)) <= 1e-8 * max(1.0, np.max(np.abs(rm)))
    assert np.max(np.abs(out[1][1][:1500] ** 2 - rs ** 2)) <= 1e-9 * C


@pytest.mark.parametrize("kid", [3, 0, 2])
def test_panel_form_follows_the_error_estimate_of_the_model(dev, kid):
    """(Matern-5/2, RBF, Matern-3/2: the three kernels that have a matrix-pipe form.)  ADVICE r04: the panel with distances from the matrix pipe carries a rounding error of up to ~4 eps C (1 + 6 R^2) per
    entry of K* (R: radius of the training set in units of the length scales), which the posterior mean multiplies by the
    weights alpha_.  The sweep estimates that product for the model at hand and takes the difference form by itself when
    it is not a factor of four inside the 1e-6 the mean is specified to: length scales at their lower bound in 16
    dimensions (R^2 = 4e6) do that -- with "cross_hybrid" = 0, "cross_mfma" = 1 and 0 then give the same bits, candidates on
    and next to training points included; with the default "cross_hybrid" = 1 such a model takes the HYBRID form (round 6:
    distances from the matrix pipe, every pair nearer than r^2 = 100 again from the coordinates), which agrees with the
    difference form to 1e-12 of the scale on exactly those candidates -- while the same data at l = 0.3 keep the matrix-pipe
    form (the two options differ in the last bits, and agree to 1e-9)."""
    N, d, M = 400, 16, 3000
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=5)
    Xc[:50] = X[:50]                                        # candidates on training points ...
    Xc[50:100] = X[50:100] + 2e-4 * (bounds[:, 1] - bounds[:, 0]) * np.random.default_rng(0).standard_normal((50, d))   # ... and a fifth of l away
    res = {}
    for name, ls in (("short", 1e-3), ("regular", 0.3)):
        m = orc.OracleGPR(bounds, kernel_id=kid)
        m.theta = np.log(np.array([3.0] + [ls] * d))
        m.fitted = True
        m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
        _load_model(dev, m)
        out, forms = {}, {}
        try:
            dev.set_option("cross_hybrid", 0)
            for mf in (1, 0):
                dev.set_option("cross_mfma", mf)
                out[mf] = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("y", "sigma"))["y"].copy()
                forms[mf] = dev.sweep_info()["panel_form"]
            dev.set_option("cross_hybrid", 1); dev.set_option("cross_mfma", 1)
            hy = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("y", "sigma"))
            out["hybrid"], out["hybrid_sigma"] = hy["y"].copy(), hy["sigma"].copy()
            forms["hybrid"] = dev.sweep_info()["panel_form"]
            dev.set_option("cross_mfma", 0)
            out["diff_sigma"] = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("y", "sigma"))["sigma"].copy()
        finally:
            dev.set_option("cross_mfma", 1); dev.set_option("cross_hybrid", 1)
        res[name] = (out, m.predict(Xc), forms)
    out, rm, forms = res["short"]
    assert forms == {1: "difference", 0: "difference", "hybrid": "hybrid"}, forms
    assert np.ptp(out[0][:100]) > 0.1                       # (the kernel does reach those candidates)
    np.testing.assert_array_equal(out[1], out[0])          # the estimate chose the difference form
    assert np.max(np.abs(out[1] - rm)) <= 1e-8 * max(1.0, np.max(np.abs(rm)))
    # the hybrid form: the near pairs come from the coordinates -- to rounding the difference form, on and next to training points too
    assert np.max(np.abs(out["hybrid"] - out[0])) <= 1e-12 * max(1.0, np.max(np.abs(rm)))
    assert np.max(np.abs(out["hybrid_sigma"] ** 2 - out["diff_sigma"] ** 2)) <= 1e-12 * np.max(out["diff_sigma"]) ** 2
    out, rm, forms = res["regular"]
    assert forms == {1: "mfma", 0: "difference", "hybrid": "mfma"}, forms
    assert not np.array_equal(out[1], out[0])               # matrix pipe: same numbers to 1e-9, not the same bits
    assert np.max(np.abs(out[1] - out[0])) <= 1e-9 * max(1.0, np.max(np.abs(rm)))
    assert np.max(np.abs(out[1] - rm)) <= 1e-8 * max(1.0, np.max(np.abs(rm)))


@pytest.mark.timeout(900)
def test_the_fitted_model_of_the_bench_against_the_oracle_and_its_panel_form():
    """the headline cycle sweeps with the model it FITTED -- bench.synthetic(4080, 16), ``fit_gpr='simple'`` from a
    random start, 16 proposals appended, another simple refit: N = 4096 with several length scales at their lower bound --, not
    with the hand-set theta = log[4, 0.3 ...] of the full-size test above.  Here that model, built exactly as bench.py builds
    it, against gpry/gpr.py:1179-1231 and gpry/gp_acquisition.py:1049-1054 through the oracle: which panel form the model's
    error estimates allowed is asserted from ``acq.stats`` (gpry_sweep_info: the variance term of round 6 rules the bare
    matrix-pipe form out for this model; its short length scales send it to the hybrid form), mean / variance / acquisition of 2048 random candidates plus 200 rows on and within l / 5 of
    training points at the suite's tolerances (1e-8 rel, 1e-9 C), the 16 proposals for identity, and the same sweep with the
    OTHER panel form forced within the north star's 1e-6 (the estimates are upper bounds, never reached on random candidates)."""
    import bench
    from gpry_amd.gp_acquisition import NORA
    n_base, d, M, npts = 4080, 16, 200_000, 16
    bounds, X, y, Xc, truth = bench.synthetic(n_base, d, M)
    gpr = bench.make_gpr(bounds)
    gpr.verbose = 0
    gpr.append_to_data(X, y, fit_gpr="simple")
    acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
    pool = {"X": Xc}
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (pool["X"], None, None, None)
    rng = np.random.default_rng(2)
    X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)
    gpr.append_to_data(X_new, truth(X_new), fit_gpr="simple")            # a timed step of the bench: N = 4096
    assert gpr.n == 4096
    theta = np.array(gpr.kernel_.theta, dtype=float)
    ls = np.exp(theta[1:])
    assert (ls < 2e-3).any(), ls                                           # (the operating point: length scales at the lower bound)
    # candidates on training points and a fifth of a length scale away from them, in front of the random pool
    span = bounds[:, 1] - bounds[:, 0]
    Xt = gpr.X_train[-100:].copy()
    near = gpr.X_train[:100] + 0.2 * ls * span * np.random.default_rng(0).standard_normal((100, d)) / np.sqrt(d)
    Xs = np.concatenate([Xt, np.clip(near, bounds[:, 0], bounds[:, 1]), Xc])
    pool["X"] = Xs
    Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(3))
    st = dict(acq.stats)
    assert st["panel_form"] in ("mfma", "difference", "hybrid") and st["panel_gate"] == 2.5e-7
    assert st["panel_error_estimate"] > 0 and st["panel_error_variance"] > 0
    assert st["panel_error_mean_worst_case"] >= st["panel_error_estimate"]
    # the gate is what decided: matrix pipe iff both estimates are inside it
    assert (st["panel_form"] == "mfma") == (st["panel_error_estimate"] <= 2.5e-7 and st["panel_error_variance"] <= 2.5e-7), st
    dev = gpr.device
    out = dev.sweep_fetch(("y", "sigma", "acq"))
    ref = orc.OracleGPR(bounds, kernel_id=orc.MATERN52)
    ref.theta = theta
    ref.fitted = True
    ref.append_to_data(gpr.X_train, gpr.y_train, fit_gpr=False, fit_preprocessors=True)
    sub = np.concatenate([np.arange(200), 200 + np.sort(np.random.default_rng(5).choice(M, 2048, replace=False))])
    rm, rs = ref.predict(Xs[sub], return_std=True)
    C = np.exp(theta[0]) * ref.pre_y.std_ ** 2
    assert np.ptp(rs[:200]) > 0 and np.min(rs[:100]) < 0.5 * np.max(rs)   # (the kernel does reach the rows next to the data)
    assert np.max(np.abs(out["y"][sub] - rm)) <= 1e-8 * np.max(np.abs(rm))
    assert np.max(np.abs(out["sigma"][sub] ** 2 - rs ** 2)) <= 1e-9 * C
    zeta = orc.auto_zeta(d)
    racq = orc.logexp_f(rm, rs, ref.y_max, ref.noise_level, zeta)
    a = out["acq"]
    assert np.array_equal(np.isneginf(a[sub]), np.isneginf(racq))
    fin = np.isfinite(racq)
    np.testing.assert_allclose(a[sub][fin], racq[fin], rtol=1e-6, atol=1e-6)
    # the proposals: the oracle's restatement of multi_add on the device shortlist united with 30 000 random rows
    order = np.lexsort((-np.arange(len(a)), -a))
    union = np.union1d(order[:max(256, st["shortlist"])], np.random.default_rng(6).choice(len(a), 30_000, replace=False))
    Xr, yr, ar = orc.nora_multi_add(ref, Xs[union], npts)
    np.testing.assert_array_equal(Xp, Xr)
    np.testing.assert_allclose(ap, ar, rtol=1e-5, atol=1e-6)
    # the other forms of the panel on the same model and rows: the difference form (and the hybrid one, whichever was not the
    # default) at the suite's tolerances; the bare matrix-pipe form, forced -- what the gate protects against, measured -- within
    # the north star's 1e-6
    try:
        for hybrid, mfma, forced, want_form, tol_m, tol_v in ((0, 0, 0, "difference", 1e-8, 1e-9), (1, 1, 0, None, 1e-8, 1e-9),
                                                              (1, 1, 32, "mfma", 1e-6, 1e-6)):
            dev.set_option("cross_hybrid", hybrid); dev.set_option("cross_mfma", mfma); dev.set_option("panel_debug", forced)
            o2 = dev.sweep_logexp(None, zeta, gpr.y_max, gpr.noise_level, M=len(Xs), want=("y", "sigma"))
            form = dev.sweep_info()["panel_form"]
            assert want_form is None or form == want_form, (form, want_form)
            assert np.max(np.abs(o2["y"][sub] - rm)) <= tol_m * max(ref.pre_y.std_, np.max(np.abs(rm))), form
            assert np.max(np.abs(o2["sigma"][sub] ** 2 - rs ** 2)) <= tol_v * C, form
    finally:
        dev.set_option("cross_mfma", 1); dev.set_option("cross_hybrid", 1)
        dev.set_option("panel_debug", 0)


@pytest.mark.parametrize("N,d,M,chunk", [(700, 5, 9000, 2048), (1500, 9, 20000, 4096)])
def test_panel_built_underneath_the_contraction_gives_the_same_sweep(dev, N, d, M, chunk):
    """Option "sweep_overlap" (round 6, measured slower and off by default: profiles/r06_sweep.md): the cross-kernel panel of
    chunk c + 1 is built on the side stream while the main stream contracts chunk c -- two panels, two sets of partial sums, one
    event per hand-over.  Same kernels on the same data: mean, sigma and acquisition (gpry/gpr.py:1179-1231,
    gpry/acquisition_functions.py:1068-1074) of every candidate are the bits of the schedule with one stream, for a resident
    pool and for one that is uploaded chunk by chunk, ragged last chunk included."""
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M - 37, seed=N)
    m = orc.OracleGPR(bounds, kernel_id=3)
    m.theta = np.log(np.array([3.0] + [0.4] * d))
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    _load_model(dev, m)
    try:
        dev.set_option("sweep_chunk", chunk)
        out = {}
        for ov in (0, 1):
            dev.set_option("sweep_overlap", ov)
            fresh = dev.sweep_logexp(Xc, 0.1, m.y_max, 1e-2, want=("y", "sigma", "acq"))
            resident = dev.sweep_logexp(None, 0.1, m.y_max, 1e-2, M=len(Xc), want=("y", "sigma", "acq"))
            for k in ("y", "sigma", "acq"):
                np.testing.assert_array_equal(fresh[k], resident[k])
            out[ov] = fresh
        for k in ("y", "sigma", "acq"):
            np.testing.assert_array_equal(out[1][k], out[0][k])
        rm, rs = m.predict(Xc[:1500], return_std=True)
        assert np.max(np.abs(out[1]["y"][:1500] - rm)) <= 1e-8 * max(1.0, np.max(np.abs(rm)))
    finally:
        dev.set_option("sweep_overlap", 0)
        dev.set_option("sweep_chunk", 0)
