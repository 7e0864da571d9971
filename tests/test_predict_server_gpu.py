"""The resident predict kernel (csrc/server.hip): mean-only ``gpry_predict`` of <= 8 points without a kernel launch
per call -- what nested samplers / MCMC call once per point (gpry/gp_acquisition.py:766-771, 784-793; gpry/mc.py:387-391).
Contract: the bits of the one-launch path (option ``predict_serve`` = 0) whatever the life cycle of the kernel did
in between (idle exit, stop by another entry point, a request posted while it was leaving), and the oracle within the
tolerance of every other predict test (mean rel <= 1e-8)."""
import threading
import time

import numpy as np
import pytest

from oracle import gpry_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _model(dev, N, d, kid, seed=0, affine=True):
    rng = np.random.default_rng(seed)
    bounds = np.array([[-2.0, 3.0]] * d)
    X = rng.uniform(bounds[:, 0], bounds[:, 1], size=(N, d))
    y = np.sin(X.sum(axis=1)) * 3.0 + 0.1 * rng.standard_normal(N)
    pre = orc.NormalizeBounds(bounds)
    X_ = pre.transform(X) if affine else X
    ym, ys = y.mean(), y.std() if N > 1 else 1.0
    theta = np.log(np.concatenate(([2.5], 0.2 + 0.3 * rng.uniform(size=d))))
    dev.set_train(X_, (y - ym) / ys, np.full(N, 1e-4))
    dev.set_theta(kid, theta)
    assert dev.factorize() == 0
    if affine:
        dev.set_affine(pre.lo, pre.hi - pre.lo, ym, ys, np.max(y) + 0.05)
    else:
        dev.set_affine(None, None, ym, ys, np.inf)
    return bounds, X_, (y - ym) / ys, theta, (ym, ys, pre)


@pytest.mark.parametrize("N,d,kid", [(1, 1, 0), (64, 2, 3), (200, 5, 1), (1000, 8, 2), (2500, 16, 3), (5000, 20, 0),
                                     (9000, 3, 3)])
def test_resident_kernel_gives_the_bits_of_the_one_launch_path_and_matches_the_oracle(N, d, kid):
    from gpry_amd import _lib
    dev = _lib.Device(0)
    try:
        bounds, X_, y_, theta, (ym, ys, pre) = _model(dev, N, d, kid, seed=N + d)
        rng = np.random.default_rng(5)
        for M in (1, 1, 2, 8, 3, 1):
            Xq = rng.uniform(bounds[:, 0], bounds[:, 1], size=(M, d))
            mask = (rng.uniform(size=M) < 0.3).astype(np.uint8) * 2 if M > 1 else None
            dev.set_option("predict_serve", 1)
            a = dev.predict(Xq, mask=mask)
            b = dev.predict(Xq, mask=mask)              # the kernel is resident now
            dev.set_option("predict_serve", 0)
            c = dev.predict(Xq, mask=mask)
            np.testing.assert_array_equal(a, c)
            np.testing.assert_array_equal(b, c)
            K = orc.kernel_matrix(pre.transform(Xq), theta, kid, Y=X_)
            L, V, alpha_ = orc.factorize(orc.kernel_matrix(X_, theta, kid) + 1e-4 * np.eye(N), y_)
            ref = np.minimum(K @ alpha_ * ys + ym, np.max(y_ * ys + ym) + 0.05)
            if mask is not None:
                ref[mask != 0] = -np.inf
            fin = np.isfinite(ref)
            assert np.array_equal(np.isfinite(c), fin)
            assert np.max(np.abs(c[fin] - ref[fin]), initial=0.0) <= 1e-8 * max(1.0, np.max(np.abs(ref[fin]), initial=0.0))
        launches, requests = dev.serve_stats()
        assert requests == 12 and 1 <= launches <= 12
    finally:
        dev.close()


def test_life_cycle_of_the_resident_kernel():
    """One launch serves thousands of calls; an entry point that changes the model stops it and the next call sees the
    NEW model; after the idle time it leaves on its own and the next call starts it again; closing the context with
    the kernel resident returns."""
    from gpry_amd import _lib
    dev = _lib.Device(0)
    try:
        bounds, X_, y_, theta, _ = _model(dev, 300, 4, 3)
        rng = np.random.default_rng(1)
        Xq = rng.uniform(bounds[:, 0], bounds[:, 1], size=(500, 4))
        dev.set_option("predict_serve", 0)
        ref = np.array([dev.predict(x[None, :])[0] for x in Xq])
        dev.set_option("predict_serve", 1)
        dev.set_option("serve_idle_us", 200000)
        t0 = time.perf_counter()
        got = np.array([dev.predict(x[None, :])[0] for x in Xq])
        dt = (time.perf_counter() - t0) / len(Xq)
        np.testing.assert_array_equal(got, ref)
        launches, requests = dev.serve_stats()
        assert (launches, requests) == (1, 500), (launches, requests)
        print(f"resident predict: {dt * 1e6:.1f} us per call through Device.predict")
        # a model change in between: new theta -> new factor -> new answers, from a new generation
        theta2 = theta + 0.3
        dev.set_theta(3, theta2)
        assert dev.factorize() == 0
        got2 = np.array([dev.predict(x[None, :])[0] for x in Xq[:50]])
        dev.set_option("predict_serve", 0)
        ref2 = np.array([dev.predict(x[None, :])[0] for x in Xq[:50]])
        np.testing.assert_array_equal(got2, ref2)
        assert not np.array_equal(ref2, ref[:50])
        dev.set_option("predict_serve", 1)
        assert dev.serve_stats() == (2, 550)
        # predictions with std, sweeps and LML evaluations in between go through their own paths
        m, s = dev.predict(Xq[:3], return_std=True)          # (its own kernels: another order of the sums)
        np.testing.assert_allclose(m, ref2[:3], rtol=1e-12, atol=1e-12)
        lml, _ = dev.lml(theta2, False)
        assert np.isfinite(lml)
        np.testing.assert_array_equal(dev.predict(Xq[:1]), ref2[:1])
        assert dev.serve_stats()[0] == 3
        # idle exit
        dev.set_option("serve_idle_us", 300)
        np.testing.assert_array_equal(dev.predict(Xq[:1]), ref2[:1])
        n0 = dev.serve_stats()[0]
        time.sleep(0.05)
        np.testing.assert_array_equal(dev.predict(Xq[1:2]), ref2[1:2])
        assert dev.serve_stats()[0] == n0 + 1
    finally:
        dev.close()        # with the kernel resident


def test_requests_that_race_with_the_idle_exit_are_served_by_the_next_generation():
    """Idle time of 30 us and host pauses scattered around it: many requests are posted while the leader is leaving.
    Every answer must be the one-launch answer (N = 3000: three workgroups, leader + followers)."""
    from gpry_amd import _lib
    dev = _lib.Device(0)
    try:
        bounds, X_, y_, theta, _ = _model(dev, 3000, 6, 3, seed=9)
        rng = np.random.default_rng(2)
        Xq = rng.uniform(bounds[:, 0], bounds[:, 1], size=(3000, 6))
        dev.set_option("predict_serve", 0)
        ref = dev.predict(Xq)                               # one batch through the panel path is NOT the same sum order
        ref = np.array([dev.predict(x[None, :])[0] for x in Xq[:200]])
        dev.set_option("predict_serve", 1)
        dev.set_option("serve_idle_us", 30)
        pauses = rng.uniform(0, 80e-6, size=len(Xq))
        bad = 0
        for i, x in enumerate(Xq):
            t_end = time.perf_counter() + pauses[i]
            while time.perf_counter() < t_end:
                pass
            v = dev.predict(x[None, :])[0]
            if i < 200:
                bad += int(v != ref[i])
        assert bad == 0
        launches, requests = dev.serve_stats()
        assert requests == len(Xq) and launches > 20, (launches, requests)      # it did leave and come back often
        print(f"{launches} generations for {requests} requests")
    finally:
        dev.close()


def test_two_contexts_with_resident_kernels_from_two_threads():
    from gpry_amd import _lib
    devs = [_lib.Device(0), _lib.Device(0)]
    try:
        out, refs = {}, {}
        models = [_model(devs[0], 500, 3, 3, seed=1), _model(devs[1], 1500, 7, 0, seed=2)]
        Xq = [np.random.default_rng(k).uniform(m[0][:, 0], m[0][:, 1], size=(400, m[0].shape[0])) for k, m in enumerate(models)]
        for k, dv in enumerate(devs):
            dv.set_option("predict_serve", 0)
            refs[k] = np.array([dv.predict(x[None, :])[0] for x in Xq[k]])
            dv.set_option("predict_serve", 1)

        def work(k):
            out[k] = np.array([devs[k].predict(x[None, :])[0] for x in Xq[k]])

        ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        for k in range(2):
            np.testing.assert_array_equal(out[k], refs[k])
    finally:
        for dv in devs:
            dv.close()


def test_point_by_point_predict_of_the_mirror_class_uses_the_resident_kernel():
    """The closure PolyChord gets (gpry/gp_acquisition.py:766-771) on the mirror regressor."""
    from test_host_mirror_gpu import make_gpr
    bounds, X, y, Xc = orc.synthetic_like_goldens(128, 4, 300, seed=4)
    theta = np.log(np.array([3.0, 0.4, 0.3, 0.5, 0.35]))
    gpr = make_gpr(bounds, 3, theta=theta)
    gpr.append_to_data(X, y, fit_gpr=False)

    def logp(x):
        return gpr.predict(np.atleast_2d(x), return_std=False, validate=False)[0]

    vals = np.array([logp(x) for x in Xc])
    ref = orc.OracleGPR(bounds, kernel_id=orc.MATERN52)
    ref.theta = theta
    ref.fitted = True
    ref.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    rm = ref.predict(Xc)
    assert np.max(np.abs(vals - rm)) <= 1e-8 * np.max(np.abs(rm))
    launches, requests = gpr.device.serve_stats()
    assert requests == len(Xc) and launches <= 3
    # appending data stops the kernel; the next calls see the enlarged model
    gpr.append_to_data(Xc[:5], np.array([logp(x) for x in Xc[:5]]) + 0.01, fit_gpr=False, fit_classifier=False)
    ref.append_to_data(Xc[:5], rm[:5] + 0.01, fit_gpr=False, fit_preprocessors=False)
    v2 = np.array([logp(x) for x in Xc[5:40]])
    r2 = ref.predict(Xc[5:40])
    assert np.max(np.abs(v2 - r2)) <= 1e-7 * np.max(np.abs(r2))


def test_classifier_and_trust_box_on_the_device_for_every_predict_path():
    """GPry's defaults (account_for_inf="SVM", a trust region): ``predict`` leaves the verdicts to the device (option
    ``predict_gates``: resident kernel for 1..8 points, gates kernel for the batches) instead of calling libsvm per
    call on the host.  Every path must give what the host-side masks give (gpry/gpr.py:1107-1112, 1145-1150 ->
    gpry/svm.py:308-347): one point at a time, small batches, with std, a few thousand points, ignore_trust_region."""
    from test_host_mirror_gpu import make_gpr
    bounds, X, y, Xc = orc.synthetic_like_goldens(150, 3, 6000, seed=9)
    y = y.copy()
    bad = X[:, 0] > 1.0
    y[bad] = -np.inf
    theta = np.log(np.array([4.0, 0.3, 0.3, 0.3]))

    def build(on_device):
        gpr = make_gpr(bounds, 3, theta=theta, account_for_inf="SVM", inf_threshold="20s", trust_region_factor=1.5,
                       random_state=1)
        gpr.append_to_data(X, y, fit_gpr=False)
        if not on_device:
            gpr.device.applies_gates_in_predict = False          # host verdicts (libsvm + numpy) as before
            gpr.device.set_option("predict_gates", 0)
        return gpr

    dev_g, host_g = build(True), build(False)
    # one point at a time: the resident kernel computes decision function and trust box itself
    for g in (dev_g, host_g):                 # (first launches load their code objects: not part of the per-call time)
        g.predict(Xc[:1], validate=False)
    t0 = time.perf_counter()
    a = np.array([dev_g.predict(x[None, :], validate=False)[0] for x in Xc[:1500]])
    t_dev = (time.perf_counter() - t0) / 1500
    t0 = time.perf_counter()
    b = np.array([host_g.predict(x[None, :], validate=False)[0] for x in Xc[:1500]])
    t_host = (time.perf_counter() - t0) / 1500
    np.testing.assert_array_equal(a, b)
    assert 50 < np.isneginf(a).sum() < 1450
    print(f"one-point predict with SVM + trust box: {t_dev * 1e6:.1f} us (device gates) vs {t_host * 1e6:.1f} us (host gates)")
    launches, requests = dev_g.device.serve_stats()
    assert requests == 1501 and launches <= 3
    # batches through the other paths
    for sl, std in ((slice(0, 5), False), (slice(0, 3), True), (slice(0, 300), True), (slice(0, 6000), True),
                    (slice(0, 6000), False), (slice(10, 17), False)):
        ra, rb = dev_g.predict(Xc[sl], return_std=std), host_g.predict(Xc[sl], return_std=std)
        if std:
            np.testing.assert_array_equal(ra[0], rb[0])
            np.testing.assert_array_equal(ra[1], rb[1])
        else:
            np.testing.assert_array_equal(ra, rb)
    # without the trust region: another set of gates on the device, same verdicts as the host's
    ra, rb = dev_g.predict(Xc[:400], ignore_trust_region=True), host_g.predict(Xc[:400], ignore_trust_region=True)
    np.testing.assert_array_equal(ra, rb)
    assert np.isneginf(ra).sum() < np.isneginf(dev_g.predict(Xc[:400])).sum()
    va = np.array([dev_g.predict(x[None, :], ignore_trust_region=True)[0] for x in Xc[:50]])
    np.testing.assert_array_equal(va, rb[:50])
    # the x-gradient branch keeps its host verdict and is not disturbed by gates held on the device
    x1 = Xc[np.flatnonzero(np.isfinite(a))[0]][None, :]
    ga, gb = dev_g.predict(x1, return_std=True, return_mean_grad=True), host_g.predict(x1, return_std=True, return_mean_grad=True)
    for u, v in zip(ga, gb):
        np.testing.assert_array_equal(u, v)
    # a refit of the classifier (new data) reaches the device
    Xn = Xc[:20]
    yn = np.where(Xn[:, 0] > 0.5, -np.inf, -1.0)
    for g in (dev_g, host_g):
        g.append_to_data(Xn, yn, fit_gpr=False)
    a2 = np.array([dev_g.predict(x[None, :], validate=False)[0] for x in Xc[100:400]])
    b2 = np.array([host_g.predict(x[None, :], validate=False)[0] for x in Xc[100:400]])
    np.testing.assert_array_equal(a2, b2)
    assert not np.array_equal(np.isneginf(a2), np.isneginf(a[100:400]))


def test_predict_std_after_a_classifier_refit_does_not_see_the_previous_gates():
    """every context applies the gates it holds inside ``gpry_predict``; ``predict_std`` (gpry/gpr.py:1275-1352:
    classifier-masked rows give 0, no trust-region gate) used to pass a host mask without syncing or clearing them, so that
    between a refit of the classifier (``append_to_data``) and the next ``predict`` the PREVIOUS support vectors still voted.
    The std of points the current classifier accepts must not be zeroed, with no ``predict`` in between."""
    from test_host_mirror_gpu import make_gpr
    bounds, X, y, Xc = orc.synthetic_like_goldens(150, 3, 600, seed=9)
    y = y.copy()
    y[X[:, 0] > 1.0] = -np.inf
    theta = np.log(np.array([4.0, 0.3, 0.3, 0.3]))

    def build(on_device):
        gpr = make_gpr(bounds, 3, theta=theta, account_for_inf="SVM", inf_threshold="20s", trust_region_factor=1.5, random_state=1)
        gpr.append_to_data(X, y, fit_gpr=False)
        if not on_device:
            gpr.device.applies_gates_in_predict = False
            gpr.device.set_option("predict_gates", 0)
        return gpr

    dev_g, host_g = build(True), build(False)
    s0d, s0h = dev_g.predict_std(Xc), host_g.predict_std(Xc)        # first call: nothing pushed yet
    np.testing.assert_array_equal(s0d, s0h)
    dev_g.predict(Xc[:300])                                           # the gates of classifier #1 now sit on the device
    # classifier #2 accepts the half-space classifier #1 rejected (the new finite points lie at x0 > 1) and rejects x1 > 1
    Xn = np.concatenate([Xc[Xc[:, 0] > 1.2][:30], Xc[Xc[:, 1] > 1.0][:30]])
    yn = np.where(Xn[:, 1] > 1.0, -np.inf, -2.0)
    for g in (dev_g, host_g):
        g.append_to_data(Xn, yn, fit_gpr=False)
    s1d, s1h = dev_g.predict_std(Xc), host_g.predict_std(Xc)        # no predict() in between
    np.testing.assert_array_equal(s1d, s1h)
    assert ((s0h == 0) != (s1h == 0)).any()                           # the two classifiers do disagree on this pool
    # a 1-D or mis-shaped X is refused by the fast path as by the general one
    with pytest.raises(ValueError):
        dev_g.device.predict(np.zeros(3))
    with pytest.raises(ValueError):
        dev_g.device.predict(np.zeros((3, 1)))
