"""Single-launch objective of small training sets (csrc/lml_small.hip: N <= 128, d <= 16; sklearn:_gpr.py:574-652 via
gpry/gpr.py:876-881) against the oracle and against the general kernel chain (option ``lml_small`` = 0).  Tolerances of
the other objective tests: LML rel <= 1e-10, gradient <= 1e-7 of its largest entry."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import gpry_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


@pytest.fixture(scope="module")
def dev():
    from gpry_amd import _lib
    d = _lib.Device(0)
    yield d
    d.close()


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("N,d", [(1, 1), (2, 3), (15, 2), (16, 4), (17, 5), (33, 8), (64, 2), (65, 16), (100, 7), (127, 3),
                                 (128, 16), (128, 1)])
def test_single_launch_objective_vs_oracle_and_general_chain(dev, kid, N, d):
    rng = np.random.default_rng(100 * N + d + kid)
    X_ = rng.uniform(size=(N, d))
    y_ = np.sin(3 * X_).sum(axis=1) + 0.05 * rng.standard_normal(N)
    y_ = (y_ - y_.mean()) / (y_.std() if N > 1 else 1.0)
    # (a one-dimensional set of 128 points with 1e-5 of noise has cond(K) ~ 1e10: there 1e-10 of the LML is the rounding
    # noise of ANY factorisation order)
    alpha = np.full(N, 1e-5 if d > 1 else 1e-3) * (1.0 + rng.uniform(size=N))
    dev.set_train(X_, y_, alpha)
    for it in range(2):
        theta = np.log(np.concatenate(([1.5 + it], 0.3 + 0.5 * rng.uniform(size=d))))
        dev.set_theta(kid, theta)
        dev.set_option("lml_small", 1)
        lml, grad, info = dev.lml(theta, True)
        lml0, info0 = dev.lml(theta, False)
        dev.set_option("lml_small", 0)
        glml, ggrad, ginfo = dev.lml(theta, True)
        dev.set_option("lml_small", 1)
        assert info == 0 and info0 == 0 and ginfo == 0
        assert lml0 == lml                               # value-only evaluation: same bits
        rl, rg = orc.log_marginal_likelihood(X_, y_, alpha, theta, kid, eval_gradient=True)
        scale = max(1.0, abs(rl))
        assert abs(lml - rl) <= 1e-10 * scale, (lml, rl)
        assert np.max(np.abs(grad - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg))), (grad, rg)
        assert abs(lml - glml) <= 1e-10 * scale
        assert np.max(np.abs(grad - ggrad)) <= 1e-7 * max(1.0, np.max(np.abs(ggrad)))


def test_single_launch_objective_on_the_reference_vectors(dev):
    """F3 goldens (values and gradients of the real reference) and its non-positive-definite convention."""
    g = load_golden("factor_lml")
    dev.set_option("lml_small", 1)
    for kid in range(4):
        X_, y_, alpha = g[f"f2_k{kid}_X_"], g[f"f2_k{kid}_y_"], g[f"f2_k{kid}_alpha"]
        th = g[f"f3_k{kid}_theta"]
        dev.set_train(X_, y_, alpha)
        dev.set_theta(kid, th)
        lml, grad, info = dev.lml(th, True)
        assert info == 0
        assert abs(lml - g[f"f3_k{kid}_lml"]) <= 1e-10 * abs(g[f"f3_k{kid}_lml"])
        np.testing.assert_allclose(grad, g[f"f3_k{kid}_grad"], rtol=1e-7, atol=1e-7 * np.max(np.abs(g[f"f3_k{kid}_grad"])))
    X_, y_, th = g["f3_nonpd_X_"], g["f3_nonpd_y_"], g["f3_nonpd_theta"]
    dev.set_train(X_, y_, np.zeros(len(y_)))
    dev.set_theta(0, th)
    outs = []
    for small in (1, 0):
        dev.set_option("lml_small", small)
        lml, grad, info = dev.lml(th, True)
        assert lml == -np.inf and not grad.any() and info > 0
        outs.append(info)
    dev.set_option("lml_small", 1)
    assert outs[0] == outs[1]                               # the same failing column as the general chain reports


def test_fit_through_the_single_launch_objective_selects_the_reference_optimum(monkeypatch):
    """F6 fits (restart schedule, L-BFGS-B) with the objective evaluated by the single-launch kernel; the prediction
    factor still comes from the general chain (gpry_factorize does not adopt anything)."""
    from test_host_mirror_gpu import make_gpr
    g = load_golden("fit")
    for kid, N in ((0, 48), (3, 60)):
        p = f"f6_k{kid}_"
        gpr = make_gpr(g[p + "bounds"], kid, n_restarts_optimizer=4, random_state=3)
        gpr.append_to_data(g[p + "X"][:N], g[p + "y"][:N], fit_gpr=True)
        assert abs(gpr.log_marginal_likelihood_value_ - g[p + "lml_full"]) < 1e-5
        np.testing.assert_allclose(gpr.kernel_.theta, g[p + "theta_full"], rtol=1e-3, atol=1e-3)
        m, s = gpr.predict(g[p + "Xc"], return_std=True)
        np.testing.assert_allclose(m, g[p + "mean_full"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(s, g[p + "std_full"], rtol=1e-4, atol=1e-5)
        assert gpr.device.timing("lml_small")[1] == 0        # (timers are off unless asked for)


@pytest.mark.gpu
@pytest.mark.parametrize("N,d,kid", [(40, 2, 0), (100, 5, 3), (128, 16, 3)])
def test_batched_objective_has_the_bits_of_single_evaluations(N, d, kid):
    """gpry_lml_batch at N <= 128: one launch, one workgroup per theta, every theta with the arithmetic of a single
    gpry_lml call -- values, gradients and the not-positive-definite verdicts equal those of B separate calls bit for bit."""
    from gpry_amd import _lib
    rng = np.random.default_rng(N + d)
    X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1) + 0.05 * rng.standard_normal(N)
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, np.full(N, 1e-8 if kid == 0 else 1e-6)); dv.set_theta(kid, np.log(np.array([2.0] + [0.4] * d)))
        thetas = np.log(np.array([2.0] + [0.4] * d)) + rng.uniform(-1.5, 1.5, (37, d + 1))
        thetas[5, 1:] = np.log(300.0)                       # a nearly singular matrix: may fail to factorise
        single = [dv.lml(th, True) for th in thetas]
        lml, grad, info = dv.lml_batch(thetas, True)
        for b, (l1, g1, i1) in enumerate(single):
            assert info[b] == i1
            assert (lml[b] == l1) or (np.isneginf(lml[b]) and np.isneginf(l1))
            np.testing.assert_array_equal(grad[b], g1)
        lml0, info0 = dv.lml_batch(thetas[:3], False)
        assert [dv.lml(th, False)[0] for th in thetas[:3]] == list(lml0)
        # larger training sets: the same entry point, evaluated one after another
        X2 = rng.uniform(size=(200, d))
        dv.set_train(X2, np.sin(3 * X2).sum(1), np.full(200, 1e-6)); dv.set_theta(kid, thetas[0])
        lml2, grad2, info2 = dv.lml_batch(thetas[:4], True)
        for b in range(4):
            l1, g1, i1 = dv.lml(thetas[b], True)
            assert lml2[b] == l1 and info2[b] == i1
            np.testing.assert_array_equal(grad2[b], g1)
    finally:
        dv.close()
