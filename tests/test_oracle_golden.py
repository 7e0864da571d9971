"""Pin the CPU oracle against vectors captured from the real reference.

The fixtures in tests/golden were written by tools/make_goldens.py driving GPry 3.0.0
itself; a pass here is what lets the GPU parity tests use the oracle as the checker.
"""
import numpy as np
import pytest

from oracle import gpry_oracle as orc
from conftest import load_golden


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b) / (np.abs(b) + 1e-300)) if a.size else 0.0


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("d", [1, 2, 5])
def test_f1_kernel_values_and_gradients(kid, d):
    g = load_golden("kernels")
    p = f"f1_k{kid}_d{d}_"
    X, Y, theta = g[p + "X"], g[p + "Y"], g[p + "theta"]
    K, dK = orc.kernel_matrix(X, theta, kid, eval_gradient=True)
    np.testing.assert_allclose(K, g[p + "K"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(dK, g[p + "dK"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(orc.kernel_matrix(Y, theta, kid, Y=X), g[p + "Kx"],
                               rtol=1e-14)
    np.testing.assert_allclose(orc.kernel_diag(Y, theta), g[p + "diag"], rtol=1e-15)


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
def test_f2_factor(kid):
    g = load_golden("factor_lml")
    p = f"f2_k{kid}_"
    K = orc.kernel_matrix(g[p + "X_"], g[p + "theta"], kid)
    K[np.diag_indices_from(K)] += g[p + "alpha"]
    L, V, a = orc.factorize(K, g[p + "y_"])
    np.testing.assert_allclose(L, g[p + "L"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(V, g[p + "V"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(a, g[p + "alpha_"], rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
def test_f3_lml_and_gradient(kid):
    g = load_golden("factor_lml")
    X_, y_, alpha = g[f"f2_k{kid}_X_"], g[f"f2_k{kid}_y_"], g[f"f2_k{kid}_alpha"]
    th = g[f"f3_k{kid}_theta"]
    lml, grad = orc.log_marginal_likelihood(X_, y_, alpha, th, kid, eval_gradient=True)
    assert abs(lml - g[f"f3_k{kid}_lml"]) <= 1e-12 * abs(g[f"f3_k{kid}_lml"])
    np.testing.assert_allclose(grad, g[f"f3_k{kid}_grad"], rtol=1e-9, atol=1e-10)
    assert orc.log_marginal_likelihood(X_, y_, alpha, th, kid) == pytest.approx(
        float(g[f"f3_k{kid}_lml_nograd"]), rel=1e-13)


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("block", [7, 256])
def test_f3_blocked_gradient_restatement_vs_reference(kid, block):
    """The memory-light form used for the full-size objective tests (N = 8192 needs 56 GB as (N, N, 1+d)
    tensors): same value and gradient as the reference's einsum (F3 goldens) and as the einsum restatement,
    on ragged row blocks and on a duplicated training row (the r = 0 branch of Matern-1/2)."""
    g = load_golden("factor_lml")
    X_, y_, alpha = g[f"f2_k{kid}_X_"], g[f"f2_k{kid}_y_"], g[f"f2_k{kid}_alpha"]
    th = g[f"f3_k{kid}_theta"]
    lml, grad = orc.log_marginal_likelihood_blocked(X_, y_, alpha, th, kid, block=block)
    assert abs(lml - g[f"f3_k{kid}_lml"]) <= 1e-12 * abs(g[f"f3_k{kid}_lml"])
    np.testing.assert_allclose(grad, g[f"f3_k{kid}_grad"], rtol=1e-9, atol=1e-10)
    rng = np.random.default_rng(kid)
    X2 = rng.uniform(size=(150, 5))
    X2[17] = X2[3]
    y2 = rng.standard_normal(150)
    th2 = np.log(np.array([3.0, 0.3, 0.5, 0.2, 0.9, 0.4]))
    a = orc.log_marginal_likelihood(X2, y2, np.full(150, 1e-3), th2, kid, eval_gradient=True)
    b = orc.log_marginal_likelihood_blocked(X2, y2, np.full(150, 1e-3), th2, kid, block=block)
    assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0])
    np.testing.assert_allclose(b[1], a[1], rtol=1e-10, atol=1e-10)
    # non-PD convention
    g3 = load_golden("factor_lml")
    lm, gr = orc.log_marginal_likelihood_blocked(g3["f3_nonpd_X_"], g3["f3_nonpd_y_"], np.zeros(len(g3["f3_nonpd_y_"])),
                                                 g3["f3_nonpd_theta"], 0)
    assert lm == -np.inf and not gr.any()


def test_f3_non_pd_convention():
    g = load_golden("factor_lml")
    X_, y_, th = g["f3_nonpd_X_"], g["f3_nonpd_y_"], g["f3_nonpd_theta"]
    assert g["f3_nonpd_lml"] == -np.inf and not g["f3_nonpd_grad"].any()
    lml, grad = orc.log_marginal_likelihood(X_, y_, np.zeros(len(y_)), th, 0, True)
    assert lml == -np.inf and not grad.any()


def _gpr_from(g, p, kid, **kw):
    gpr = orc.OracleGPR(g[p + "bounds"], kernel_id=kid, **kw)
    gpr.theta = np.array(g[p + "theta"])
    gpr.fitted = True
    return gpr


@pytest.mark.parametrize("kid", [0, 3])
def test_f4_predict_clip_and_trust_region(kid):
    g = load_golden("predict")
    p = f"f4_k{kid}_"
    gpr = _gpr_from(g, p, kid, clip_factor=1.0)
    gpr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False, fit_preprocessors=True)
    mean, std = gpr.predict(g[p + "Xc"], return_std=True)
    assert (g[p + "mean"] == g[p + "clip_hi"]).sum() >= 1  # clipping exercised
    np.testing.assert_allclose(mean, g[p + "mean"], rtol=1e-10, atol=1e-10)
    C = np.exp(g[p + "theta"][0]) * gpr.pre_y.std_ ** 2
    assert np.max(np.abs(std ** 2 - g[p + "std"] ** 2)) <= 1e-10 * C
    np.testing.assert_allclose(gpr.predict_std(g[p + "Xc"]), g[p + "std_only"],
                               rtol=1e-7, atol=1e-9)
    gtr = _gpr_from(g, p, kid, trust_region_factor=0.6, clip_factor=1.0)
    gtr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False, fit_preprocessors=True)
    np.testing.assert_allclose(gtr.trust_bounds, g[p + "trust_bounds"], rtol=1e-14)
    mean_tr, std_tr = gtr.predict(g[p + "Xc"], return_std=True)
    assert np.array_equal(np.isneginf(mean_tr), np.isneginf(g[p + "mean_tr"]))
    assert np.isneginf(mean_tr).sum() > 0
    fin = np.isfinite(mean_tr)
    np.testing.assert_allclose(mean_tr[fin], g[p + "mean_tr"][fin], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(std_tr, g[p + "std_tr"], rtol=1e-7, atol=1e-9)


def test_f4_classifier_mask_semantics():
    """With a classifier: masked rows get mean=-inf, std=0 (gpry/gpr.py:1145,1172,1230)."""
    g = load_golden("predict")
    p = "f4_svm_"
    gpr = _gpr_from(g, p, 3)
    gpr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False, fit_preprocessors=True)
    finite = g[p + "finite"]
    assert 0 < finite.sum() < len(finite)
    mean, std = gpr.predict(g[p + "Xc"], return_std=True)
    mean = np.where(finite, mean, -np.inf)
    std = np.where(finite, std, 0.0)
    assert np.array_equal(np.isneginf(mean), np.isneginf(g[p + "mean"]))
    np.testing.assert_allclose(mean[finite], g[p + "mean"][finite], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(std, g[p + "std"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(std, g[p + "std_only"], rtol=1e-7, atol=1e-9)


def test_f5_logexp():
    g = load_golden("predict")
    acq = orc.logexp_f(g["f5_mu"], g["f5_std"], float(g["f5_baseline"]),
                       float(g["f5_noise"]), float(g["f5_zeta"]))
    assert np.array_equal(np.isneginf(acq), np.isneginf(g["f5_acq"]))
    assert np.isneginf(acq).sum() >= 4
    fin = np.isfinite(acq)
    np.testing.assert_array_equal(acq[fin], g["f5_acq"][fin])
    assert float(g["f5_zeta"]) == orc.auto_zeta(7)


@pytest.mark.parametrize("kid,N", [(0, 48), (3, 60)])
def test_f6_fit_full_and_simple(kid, N):
    g = load_golden("fit")
    p = f"f6_k{kid}_"
    gpr = orc.OracleGPR(g[p + "bounds"], kernel_id=kid, n_restarts_optimizer=4,
                        random_state=3)
    np.testing.assert_allclose(gpr.theta_bounds, g[p + "theta_bounds"], rtol=1e-15)
    X, y, Xc = g[p + "X"], g[p + "y"], g[p + "Xc"]
    gpr.append_to_data(X[:N], y[:N], fit_gpr=True)
    # L-BFGS-B trajectories are sensitive to 1e-13 objective differences (flat optimum):
    # compare fits by final LML and by predictions, not by theta or call counts.
    assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_full"]) < 1e-5
    np.testing.assert_allclose(gpr.theta, g[p + "theta_full"], rtol=1e-3, atol=1e-3)
    m, s = gpr.predict(Xc, return_std=True)
    np.testing.assert_allclose(m, g[p + "mean_full"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s, g[p + "std_full"], rtol=1e-4, atol=1e-5)
    gpr.append_to_data(X[N:], y[N:], fit_gpr="simple")
    assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_simple"]) < 1e-4
    m, s = gpr.predict(Xc, return_std=True)
    np.testing.assert_allclose(m, g[p + "mean_simple"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(s, g[p + "std_simple"], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("kid", [3, 0])
def test_f6b_fit_of_a_few_hundred_points(kid):
    """F6b (tools/make_goldens.py:f6b_fit_mid): the reference's 4-restart fits of 200 / 300 points -- the sizes at which the
    device steps the restarts side by side on the batched chain (tests/test_lml_batch_gpu.py)."""
    g = load_golden("fit_mid")
    p = f"f6b_k{kid}_"
    gpr = orc.OracleGPR(g[p + "bounds"], kernel_id=kid, n_restarts_optimizer=4, random_state=3)
    gpr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=True)
    assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_full"]) < 1e-5 * max(1.0, abs(float(g[p + "lml_full"])))
    np.testing.assert_allclose(gpr.theta, g[p + "theta_full"], rtol=1e-3, atol=1e-3)
    m, s = gpr.predict(g[p + "Xc"], return_std=True)
    np.testing.assert_allclose(m, g[p + "mean_full"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s, g[p + "std_full"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_f7_nora_multi_add(tag):
    g = load_golden("multi_add")
    p = f"f7{tag}_"
    kid, M = int(g[p + "kid"]), int(g[p + "M"])
    N, d = g[p + "X"].shape
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, int(g[p + "seed"]))
    np.testing.assert_array_equal(X, g[p + "X"])
    np.testing.assert_allclose(Xc.sum(axis=0), g[p + "Xc_sum"], rtol=0, atol=0)
    gpr = orc.OracleGPR(bounds, kernel_id=kid)
    gpr.theta = np.array(g[p + "theta"])
    gpr.fitted = True
    gpr.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    npts = len(g[p + "acq_cond"]) - 1
    Xp, yp, ap, info = orc.nora_multi_add(gpr, Xc, npts, return_all=True)
    np.testing.assert_allclose(info["y"], g[p + "y_mc"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(info["sigma"], g[p + "sigma_mc"], rtol=1e-7, atol=1e-9)
    np.testing.assert_array_equal(Xp, g[p + "X_pool"])  # identical argmax / ranking
    np.testing.assert_allclose(yp, g[p + "y_pool"], rtol=1e-10)
    np.testing.assert_allclose(ap, g[p + "acq_pool"], rtol=1e-8)
    np.testing.assert_allclose(info["acq_cond"], g[p + "acq_cond"], rtol=1e-6)
    assert info["cache_counter"] == int(g[p + "cache_counter"])
    # second call: absorbed first batch, re-used sample minus already-proposed rows
    gpr.append_to_data(Xp, g[p + "y_new"], fit_gpr=False, fit_preprocessors=True)
    Xp2, yp2, ap2 = orc.nora_multi_add(gpr, Xc, npts, already_proposed=Xp)
    np.testing.assert_array_equal(Xp2, g[p + "X_pool2"])
    np.testing.assert_allclose(yp2, g[p + "y_pool2"], rtol=1e-9)
    np.testing.assert_allclose(ap2, g[p + "acq_pool2"], rtol=1e-7)


def test_f8_append_rows_fixed_theta():
    g = load_golden("predict")
    gpr = _gpr_from(g, "f8_", 2)
    X, y, Xc = g["f8_X"], g["f8_y"], g["f8_Xc"]
    gpr.append_to_data(X[:32], y[:32], fit_gpr=False, fit_preprocessors=True)
    np.testing.assert_allclose(gpr.predict_std(Xc), g["f8_std_before"], rtol=1e-8)
    gpr.append_to_data(X[32:], y[32:], fit_gpr=False, fit_preprocessors=False)
    np.testing.assert_allclose(gpr.X_train_, g["f8_X_train_"], rtol=1e-15)
    np.testing.assert_allclose(gpr.y_train_, g["f8_y_train_"], rtol=1e-13)
    np.testing.assert_allclose(gpr.L_, g["f8_L"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(gpr.V_, g["f8_V"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(gpr.alpha_, g["f8_alpha_"], rtol=1e-9, atol=1e-10)
    m, s = gpr.predict(Xc, return_std=True)
    np.testing.assert_allclose(m, g["f8_mean_after"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(s, g["f8_std_after"], rtol=1e-7, atol=1e-9)


def test_f9_config1_curved_degeneracy_plumbing():
    g = load_golden("fit")
    X, y = g["f9_X"], g["f9_y"]
    np.testing.assert_allclose(orc.curved_degeneracy(X), y, rtol=1e-15)
    np.testing.assert_array_equal(orc.CURVED_BOUNDS, g["f9_bounds"])
    gpr = orc.OracleGPR(g["f9_bounds"], kernel_id=0, n_restarts_optimizer=3,
                        random_state=3)
    gpr.append_to_data(X, y, fit_gpr=True)
    assert abs(gpr.log_marginal_likelihood_value_ - g["f9_lml"]) < 1e-6
    m, s = gpr.predict(g["f9_Xc"], return_std=True)
    np.testing.assert_allclose(m, g["f9_mean"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s, g["f9_std"], rtol=1e-4, atol=1e-6)


# ---- F10: x-gradients ------------------------------------------
@pytest.mark.parametrize("kid", [0, 1, 2, 3])
def test_f10_kernel_gradient_x_vs_reference(kid):
    g = load_golden("gradients")
    pre = orc.NormalizeBounds(g["f10_bounds"])
    X_ = pre.transform(g["f10_X"])
    theta = g[f"f10_k{kid}_theta"]
    for i, x in enumerate(g["f10_Xc"]):
        got = orc.kernel_gradient_x(pre.transform(x[None, :])[0], X_, theta, kid)
        np.testing.assert_allclose(got, g[f"f10_k{kid}_kgrad"][i], rtol=1e-11, atol=1e-13)
    assert int(g["f10_k1_reference_raises"]) == 1   # the reference's Matern-1/2 path is broken


@pytest.mark.parametrize("kid", [0, 2, 3])
def test_f10_predict_gradients_and_logexp_gradient_vs_reference(kid):
    g = load_golden("gradients")
    p = f"f10_k{kid}_"
    m = orc.OracleGPR(g["f10_bounds"], kernel_id=kid)
    m.theta = np.array(g[p + "theta"])
    m.fitted = True
    m.append_to_data(g["f10_X"], g["f10_y"], fit_gpr=False, fit_preprocessors=True)
    for i, x in enumerate(g["f10_Xc"]):
        mean, std, mg, sg = m.predict_with_grad(x)
        np.testing.assert_allclose(mean[0], g[p + "mean"][i], rtol=1e-9)
        np.testing.assert_allclose(std[0], g[p + "std"][i], rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(mg, g[p + "mean_grad"][i], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sg, g[p + "std_grad"][i], rtol=1e-5, atol=1e-8)
        ag = orc.logexp_gradient(std[0], mg, sg, m.noise_level, float(g[p + "zeta"]))
        ref = g[p + "acq_grad"][i]
        assert np.array_equal(np.isinf(ag), np.isinf(ref))
        fin = np.isfinite(ref)
        np.testing.assert_allclose(ag[fin], ref[fin], rtol=1e-5, atol=1e-7)
