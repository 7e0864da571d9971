"""GPU tests of the single-process device group (``gpry_group_*`` / ``NORA(devices=...)``): k contexts
on device 0 driven from k host threads shard the candidate pool exactly as k ranks would
(BASELINE configs[3] logic on a 1-GPU box).  Everything must equal the one-context result and the
reference's golden vectors (gpry/gp_acquisition.py:971-1191, gpry/mpi.py:105-131)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import gpry_oracle as orc
from test_host_mirror_gpu import make_gpr

pytestmark = pytest.mark.gpu


def _f7(tag):
    g = load_golden("multi_add")
    p = f"f7{tag}_"
    kid, M = int(g[p + "kid"]), int(g[p + "M"])
    N, d = g[p + "X"].shape
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, int(g[p + "seed"]))
    gpr = make_gpr(bounds, kid, theta=g[p + "theta"])
    gpr.append_to_data(X, y, fit_gpr=False)
    return g, p, bounds, Xc, gpr


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("k", [2, 3, 8])
def test_f7_multi_add_sharded_over_k_contexts_equals_one_context_and_the_reference(tag, k):
    from gpry_amd.gp_acquisition import NORA
    g, p, bounds, Xc, gpr = _f7(tag)
    npts = len(g[p + "acq_cond"]) - 1
    res = {}
    for name, devices in (("one", [0]), ("group", [0] * k)):
        if name == "group":
            g2, _, _, _, gpr = _f7(tag)            # fresh model: same state as the first run started from
        acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, shortlist_size=8, devices=devices)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
        first = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        cond1 = acq.pool.acq_cond.copy()
        sample1 = [np.copy(v) for v in acq.last_MC_sample(warn_reweight=False)[:3]]
        gpr.append_to_data(first[0], g[p + "y_new"], fit_gpr=False)
        second = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))   # reweighted sample
        sample2 = acq.last_MC_sample(warn_reweight=False)
        res[name] = (first, cond1, sample1, second, acq.pool.acq_cond.copy(), sample2, acq.stats["sweep_contexts"])
    one, grp = res["one"], res["group"]
    assert one[6] == 1 and grp[6] == k
    for a, b in zip(one[0], grp[0]):
        np.testing.assert_array_equal(a, b)            # proposals, lies, acquisition: bit-identical
    np.testing.assert_array_equal(one[1], grp[1])      # conditioned acquisition of the pool
    for a, b in zip(one[2], grp[2]):
        np.testing.assert_array_equal(a, b)            # y / sigma of the whole pool
    for a, b in zip(one[3], grp[3]):
        np.testing.assert_array_equal(a, b)            # second call (reweighting, exclusions)
    np.testing.assert_array_equal(one[4], grp[4])
    for a, b in zip(one[5], grp[5]):
        np.testing.assert_array_equal(a, b)
    # and against the reference's vectors
    np.testing.assert_array_equal(grp[0][0], g[p + "X_pool"])
    np.testing.assert_allclose(grp[0][2], g[p + "acq_pool"], rtol=1e-7)
    np.testing.assert_allclose(grp[1], g[p + "acq_cond"], rtol=1e-5)
    np.testing.assert_array_equal(grp[3][0], g[p + "X_pool2"])
    np.testing.assert_allclose(grp[3][2], g[p + "acq_pool2"], rtol=1e-6)
    assert len(grp[5][1]) == int(g[p + "n_rw"])


@pytest.mark.parametrize("k", [2, 5])
def test_group_shortlist_is_the_prefix_of_the_global_descending_stream(k):
    """gpry_group_sweep_topk against np.lexsort on the whole pool: merged order, hold-back rule,
    exclusions given as global rows, ragged shards (M not divisible by k), ties, exhaustion."""
    from gpry_amd import _lib
    bounds, X, y, Xc = orc.synthetic_like_goldens(96, 3, 5003, seed=9)
    Xc[700:712] = Xc[3]                     # ties across what will be different shards
    Xc[4000:4004] = Xc[3]
    gpr = make_gpr(bounds, 3, theta=np.log(np.array([4.0, 0.3, 0.3, 0.3])))
    gpr.append_to_data(X, y, fit_gpr=False)
    gpr._ensure_factor()
    gpr._push_affine()
    grp = _lib.DeviceGroup([0] * k, adopt=gpr.device)
    assert grp.size == k and grp.transport == "host"        # one device: no RCCL between the members
    kid, theta = gpr._device_theta()
    assert grp.set_model(gpr.X_train_, gpr.y_train_, gpr.alpha, kid, theta, gpr._affine_args()) == 0
    ref = gpr.device.sweep_logexp(Xc, 0.4, gpr.y_max, gpr.noise_level)    # one context, whole pool
    out = grp.sweep_logexp(Xc, 0.4, gpr.y_max, gpr.noise_level)           # (member 0 IS that context)
    for key in ("y", "sigma", "acq"):
        np.testing.assert_array_equal(out[key], ref[key])
    acq = out["acq"]
    M = len(acq)
    order = np.lexsort((-np.arange(M), -acq))
    per = -(-M // k)
    for K in (1, 5, 64, 1500, 6000):
        top, bound, exhausted = grp.sweep_topk(K)
        n = len(top)
        np.testing.assert_array_equal(top["idx"], order[:n])        # a prefix of the global stream
        np.testing.assert_array_equal(top["acq"], acq[order[:n]])
        np.testing.assert_array_equal(top["y"], out["y"][order[:n]])
        assert exhausted == (K > per) or (exhausted and K >= per)
        if exhausted:
            assert n == M and bound == -np.inf
        else:
            assert n >= min(K, M)                                    # at least one member's K survive
            assert np.all(acq[order[n:]] <= bound) and np.all(top["acq"] > bound)
    excl = np.sort(order[[0, 1, 5, 40, 41, 977]])
    top, bound, _ = grp.sweep_topk(50, exclude=excl)
    keep = order[~np.isin(order, excl)]
    np.testing.assert_array_equal(top["idx"], keep[:len(top)])
    # resident shards: X=None re-runs the same sweep
    again = grp.sweep_logexp(None, 0.4, gpr.y_max, gpr.noise_level, M=M)
    np.testing.assert_array_equal(again["acq"], acq)
    fetched = grp.sweep_fetch(("y", "sigma", "acq"))
    np.testing.assert_array_equal(fetched["sigma"], out["sigma"])
    with pytest.raises(_lib.GpryHipError):
        grp.sweep_logexp(None, 0.4, gpr.y_max, gpr.noise_level, M=M - 1)


def test_group_lml_batch_equals_sequential_evaluations():
    from gpry_amd import _lib
    g = load_golden("fit")
    p = "f6_k3_"
    gpr = make_gpr(g[p + "bounds"], 3, theta=np.asarray(g[p + "theta_full"]))
    gpr.append_to_data(g[p + "X"], g[p + "y"], fit_gpr=False)
    gpr._ensure_factor()
    grp = _lib.DeviceGroup([0, 0, 0], adopt=gpr.device)
    kid, theta = gpr._device_theta()
    grp.set_model(gpr.X_train_, gpr.y_train_, gpr.alpha, kid, theta, gpr._affine_args())
    rng = np.random.default_rng(4)
    thetas = theta + 0.3 * rng.standard_normal((7, len(theta)))
    lml, grad, info = grp.lml_batch(thetas, True)
    for t in range(7):
        v, gr, inf = gpr.device.lml(thetas[t], True)
        assert v == lml[t] and inf == info[t]
        np.testing.assert_array_equal(gr, grad[t])
    # the prediction factor of the adopted context survived the evaluations
    m1 = gpr.predict(g[p + "Xc"][:8])
    gpr._invalidate()
    np.testing.assert_array_equal(m1, gpr.predict(g[p + "Xc"][:8]))


def test_group_applies_device_gates_on_every_member():
    from gpry_amd.gp_acquisition import NORA
    bounds, X, y, Xc = orc.synthetic_like_goldens(200, 4, 30000, seed=31)
    y = y.copy()
    y[X[:, 0] > 1.0] = -np.inf
    res = []
    for devices in ([0], [0, 0, 0]):
        gpr = make_gpr(bounds, 3, theta=np.log(np.array([4.0, 0.3, 0.3, 0.3, 0.3])), account_for_inf="SVM",
                       inf_threshold="20s", trust_region_factor=1.5, random_state=1)
        gpr.append_to_data(X, y, fit_gpr=False)
        acq = NORA(bounds, sampler="uniform", verbose=0, devices=devices, mc_every=1)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
        out = acq.multi_add(gpr, n_points=4, bounds=gpr.trust_bounds, rng=np.random.default_rng(0))
        res.append(out + tuple(acq.last_MC_sample()[1:3]))
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)
    assert np.isneginf(res[1][3]).sum() > 100           # the gates did mask candidates


@pytest.mark.timeout(900)
def test_config3_at_full_size_sharded_eight_ways_equals_one_context():
    """BASELINE configs[3] logic at full size on one GPU: N=4096, d=16, M=1e6 candidates sharded over 8 contexts
    (125 000 candidates = 4 launches each, model replicated and factorised on every member, shortlists merged
    with the hold-back rule): proposals, lies, acquisition values, the ranked pool and the y / sigma arrays of
    all 1e6 candidates must equal the one-context run bit for bit.

    This is a SELF-comparison (8 contexts against 1): it pins the sharding, not the arithmetic.  The one-context run
    at this very size is anchored to the oracle by
    ``tests/test_hip_parity.py::test_full_size_config2_sweep_topk_and_multi_add_vs_oracle`` (same N, d, M, theta,
    pool: shortlist = head of np.lexsort over all 1e6 acquisition values, 2048 rows and the 16 proposals against
    the oracle), so the sharded result reaches the oracle through that test."""
    import bench
    from gpry_amd.gp_acquisition import NORA
    N, d, M, npts = 4096, 16, 1_000_000, 16
    bounds, X, y, Xc, _ = bench.synthetic(N, d, M)
    theta = np.log(np.array([4.0] + [0.3] * d))
    res = {}
    for name, devices in (("one", [0]), ("eight", [0] * 8)):
        gpr = make_gpr(bounds, 3, theta=theta, noise_level=1e-2)
        gpr.append_to_data(X, y, fit_gpr=False)
        acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=devices)
        acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
        out = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
        sample = acq.last_MC_sample()
        res[name] = (out, acq.pool.acq_cond.copy(), sample[1], sample[2], acq.stats["sweep_contexts"])
    assert res["one"][4] == 1 and res["eight"][4] == 8
    for a, b in zip(res["one"][0], res["eight"][0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res["one"][1], res["eight"][1])
    np.testing.assert_array_equal(res["one"][2], res["eight"][2])
    np.testing.assert_array_equal(res["one"][3], res["eight"][3])
