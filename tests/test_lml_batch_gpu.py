"""The batched objective above N = 128 (``gpry_lml_batch`` -> ``lml_batch_general``, csrc/api.hip): the thetas of a call go
through ONE chain of launches, every kernel of an evaluation carrying all of them (grid.z) with a scratch set per theta.
Reference: sklearn:_gpr.py:574-652 via gpry/gpr.py:876-881; the restarts that produce such batches run one after another
in gpry/gpr.py:968-984.  Bar: per theta the BITS of a single ``gpry_lml`` (same kernels, same launch geometry, same
operands), the oracle's value (rel <= 1e-10) and gradient (<= 1e-7 of its largest entry), a not-positive-definite theta
reported on its own."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import gpry_oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def _problem(N, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, d))
    y = np.sin(3 * X).sum(1) + 0.05 * rng.standard_normal(N)
    y = (y - y.mean()) / y.std()
    alpha = 1e-5 * (1.0 + rng.uniform(size=N))
    return rng, X, y, alpha


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("N,d,B", [(129, 3, 7), (300, 5, 12), (1000, 8, 9), (2048, 16, 5)])
def test_batched_chain_has_the_bits_of_single_evaluations_and_the_oracle_values(N, d, B, kid):
    from gpry_amd import _lib
    rng, X, y, alpha = _problem(N, d, 7 * N + d + kid)
    base = np.log(np.array([2.0] + [0.5] * d))
    thetas = base + rng.uniform(-0.7, 0.7, (B, d + 1))
    # One theta whose matrix cannot be factorised: six training points are duplicated, which every other theta survives
    # thanks to the noise on their diagonal entries (pivot ~ 2 alpha = 0.02: the matrices stay well conditioned, so that the
    # oracle comparison below keeps the tolerances of the other objective tests), while at C = 1e15 that noise is below
    # half an ulp of the diagonal: the duplicated rows of K are identical and their pivots are rounding noise around zero.
    bad = 2
    for j in range(0, 12, 2):
        X[j + 1] = X[j]
    alpha[:12] = 1e-2
    thetas[bad, 0] = np.log(1e15)
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, alpha)
        dv.set_theta(kid, base)
        assert dv.lml_batch_max >= 2048
        single = [dv.lml(th, True) for th in thetas]
        lml, grad, info = dv.lml_batch(thetas, True)
        for b, (l1, g1, i1) in enumerate(single):
            assert info[b] == i1, (b, info[b], i1)
            assert lml[b] == l1 or (np.isneginf(lml[b]) and np.isneginf(l1)), (b, lml[b], l1)
            np.testing.assert_array_equal(grad[b], g1)
        assert info[bad] > 0 and np.isneginf(lml[bad]) and not grad[bad].any()
        assert all(info[b] == 0 for b in range(B) if b != bad)
        # value-only batch: the bits of value-only single calls, and of the values above
        lml0, info0 = dv.lml_batch(thetas, False)
        for b in range(B):
            assert info0[b] == info[b]
            assert lml0[b] == lml[b] or (np.isneginf(lml0[b]) and np.isneginf(lml[b]))
        # the oracle (the einsum restatement is O(N^2 d) memory: two thetas at the large sizes)
        for b in ([0, 1, 3] if N <= 1000 else [0, 1]):
            if info[b]:
                continue
            rl, rg = orc.log_marginal_likelihood(X, y, alpha, thetas[b], kid, eval_gradient=True)
            assert abs(lml[b] - rl) <= 1e-10 * max(1.0, abs(rl)), (lml[b], rl)
            assert np.max(np.abs(grad[b] - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg))), (grad[b], rg)
        # the prediction factor is not disturbed by a batch: factorise, batch, predict = factorise, predict
        info_f = dv.factorize()
        if info_f == 0:
            Xc = rng.uniform(size=(64, d))
            m0, s0 = dv.predict(Xc, return_std=True)
            dv.lml_batch(thetas, True)
            m1, s1 = dv.predict(Xc, return_std=True)
            np.testing.assert_array_equal(m0, m1)
            np.testing.assert_array_equal(s0, s1)
    finally:
        dv.close()


@pytest.mark.parametrize("N,d,B,kid", [(4096, 16, 5, 3), (5000, 6, 4, 0), (7300, 8, 3, 3)])
def test_batched_chain_at_the_sizes_of_the_pipelined_and_the_separate_launch_schedules(N, d, B, kid):
    """A single evaluation queues V = L^-1 underneath potrf on a second stream, and above Np = 3584 the Cholesky runs its
    first columns in outer blocks with a SYRK launch behind each: a batch keeps everything on the main stream (it has thetas enough
    to fill the GPU) and carries the thetas through the SYRK launches as well -- the per-theta bits are still those of the
    single evaluations (the schedules are bit-identical), and the values are the oracle's at the sizes its blocked
    restatement reaches in seconds."""
    from gpry_amd import _lib
    rng, X, y, alpha = _problem(N, d, N + d)
    base = np.log(np.array([2.0] + [0.5] * d))
    thetas = base + rng.uniform(-0.4, 0.4, (B, d + 1))
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, alpha)
        dv.set_theta(kid, base)
        dv.set_option("lml_batch", 8192)        # (the default stops at 4096, where the host's thread farm takes over)
        single = [dv.lml(th, True) for th in thetas]
        lml, grad, info = dv.lml_batch(thetas, True)
        for b, (l1, g1, i1) in enumerate(single):
            assert info[b] == i1 == 0
            assert lml[b] == l1, (b, lml[b], l1)
            np.testing.assert_array_equal(grad[b], g1)
        if N <= 4096:
            rl, rg = orc.log_marginal_likelihood_blocked(X, y, alpha, thetas[0], kid)     # (memory-light restatement, pinned on F3)
            assert abs(lml[0] - rl) <= 1e-10 * max(1.0, abs(rl))
            assert np.max(np.abs(grad[0] - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg)))
    finally:
        dv.close()


def test_batches_longer_than_the_arena_go_through_in_chunks():
    """``lml_batch_mb`` caps the scratch arena: a batch that does not fit is cut into chunks (and one that cannot hold two
    sets is evaluated theta by theta) -- same bits either way."""
    from gpry_amd import _lib
    rng, X, y, alpha = _problem(640, 4, 5)
    thetas = np.log(np.array([2.0] + [0.5] * 4)) + rng.uniform(-0.5, 0.5, (11, 5))
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, alpha); dv.set_theta(3, thetas[0])
        ref = dv.lml_batch(thetas, True)
        for mb in (64, 8, 1):                  # a set at Np = 640 is ~13-17 MB: 64 MiB -> chunks of 3-4; 8 MiB -> sequential
            dv.set_option("lml_batch_mb", mb)
            out = dv.lml_batch(thetas, True)
            for a, b in zip(ref, out):
                np.testing.assert_array_equal(a, b)
        dv.set_option("lml_batch_mb", 49152)
        dv.set_option("lml_batch", 0)           # off: one after another
        out = dv.lml_batch(thetas, True)
        for a, b in zip(ref, out):
            np.testing.assert_array_equal(a, b)
        # 130 thetas: more than one chain carries (96)
        many = np.repeat(thetas, 12, axis=0)[:130]
        dv.set_option("lml_batch", 4096)
        lm, gr, inf = dv.lml_batch(many, True)
        for b in range(130):
            assert lm[b] == ref[0][b // 12] and inf[b] == 0
            np.testing.assert_array_equal(gr[b], ref[1][b // 12])
    finally:
        dv.close()


def test_a_batch_after_a_change_of_the_training_set_and_between_sizes():
    """The arena is laid out per call: growing / shrinking training sets, another dimension, another kernel."""
    from gpry_amd import _lib
    dv = _lib.Device(0)
    try:
        for N, d, kid in ((200, 2, 3), (900, 6, 0), (260, 6, 2), (1500, 3, 1), (130, 20, 3)):
            rng, X, y, alpha = _problem(N, d, N + d)
            thetas = np.log(np.array([1.5] + [0.6] * d)) + rng.uniform(-0.4, 0.4, (6, d + 1))
            dv.set_train(X, y, alpha); dv.set_theta(kid, thetas[0])
            lml, grad, info = dv.lml_batch(thetas, True)
            for b in range(6):
                l1, g1, i1 = dv.lml(thetas[b], True)
                assert lml[b] == l1 and info[b] == i1 == 0
                np.testing.assert_array_equal(grad[b], g1)
    finally:
        dv.close()


@pytest.mark.parametrize("kid", [3, 0])
def test_side_by_side_fit_above_128_points_equals_the_sequential_fit_and_the_reference(kid, monkeypatch):
    """The default multi-restart fit of a few hundred points steps its restarts together (``_restarts_side_by_side``) on the
    batched chain: every run sees the bits it would see alone, so the selected theta, the LML and the evaluation count
    equal those of the sequential loop bit for bit; the F6b goldens (the reference's own 4-restart fits of 200 / 300
    points, tools/make_goldens.py:f6b_fit_mid) pin both within the tolerances of the F6 test."""
    from test_host_mirror_gpu import make_gpr
    g = load_golden("fit_mid")
    p = f"f6b_k{kid}_"
    X, y, Xc = g[p + "X"], g[p + "y"], g[p + "Xc"]
    assert len(X) > 128
    res = {}
    for lock in ("1", "0"):
        monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", lock)
        monkeypatch.setenv("GPRY_HIP_FIT_CONTEXTS", "1")
        gpr = make_gpr(g[p + "bounds"], kid, n_restarts_optimizer=4, random_state=3)
        gpr.append_to_data(X, y, fit_gpr=True)
        res[lock] = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike,
                     dict(getattr(gpr, "fit_stats", None) or {}), gpr.predict(Xc, return_std=True))
    assert res["1"][3].get("side_by_side") is True and not res["0"][3].get("side_by_side")
    np.testing.assert_array_equal(res["1"][0], res["0"][0])
    assert res["1"][1] == res["0"][1]
    assert res["1"][2] == res["0"][2]
    np.testing.assert_array_equal(res["1"][4][0], res["0"][4][0])
    np.testing.assert_array_equal(res["1"][4][1], res["0"][4][1])
    theta, lml, neval, _, (m, s) = res["1"]
    assert abs(lml - g[p + "lml_full"]) < 1e-5 * max(1.0, abs(float(g[p + "lml_full"])))
    np.testing.assert_allclose(theta, g[p + "theta_full"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(m, g[p + "mean_full"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s, g[p + "std_full"], rtol=1e-4, atol=1e-5)
    # (the evaluation count is compared with the sequential loop's above, not with the reference's: L-BFGS-B trajectories
    # amplify 1e-13 differences of the objective -- 205 vs 284 evaluations for the same optimum here)
    assert neval > 10


def test_side_by_side_runs_in_independent_groups_on_one_gpu(monkeypatch):
    """The runs of a side-by-side fit dealt out over two / three independent groups (own lock-step driver, host thread and
    context on the same GPU: their kernel chains run beside each other): same optimum, LML and evaluation count as with
    one group, bit for bit."""
    from test_host_mirror_gpu import make_gpr
    bounds, X, y, _ = orc.synthetic_like_goldens(900, 5, 8, seed=12)
    out = {}
    for k in ("3", "2", "1"):
        monkeypatch.setenv("GPRY_HIP_FIT_BATCH_CONTEXTS", k)
        monkeypatch.setenv("GPRY_HIP_FIT_LOCKSTEP", "1")
        gpr = make_gpr(bounds, 3, n_restarts_optimizer=9, random_state=4)
        gpr.fit_devices = [0, 0, 0]
        gpr.append_to_data(X, y, fit_gpr=True)
        assert gpr.fit_stats["side_by_side"] and gpr.fit_stats["contexts"] == int(k)
        out[k] = (gpr.kernel_.theta.copy(), gpr.log_marginal_likelihood_value_, gpr.n_eval_loglike)
    for k in ("3", "2"):
        np.testing.assert_array_equal(out[k][0], out["1"][0])
        assert out[k][1] == out["1"][1] and out[k][2] == out["1"][2]


# ---- the throughput schedule (option "lml_schedule" = 1, round 6) ---------------------------------------------------------------
@pytest.mark.parametrize("kid", [0, 3])
@pytest.mark.parametrize("N,d,B", [(129, 3, 7), (300, 5, 12), (1000, 8, 9), (1100, 6, 10), (2048, 16, 6), (2600, 12, 5)])
def test_throughput_schedule_is_independent_of_the_batch_and_agrees_with_the_oracle(N, d, B, kid):
    """``lml_schedule`` = 1: the chain for many thetas at once -- whole-tile products only (no partial slices), the recursive
    inverse at every size, the Cholesky in column blocks with a SYRK launch behind each, the thetas dealt over stream groups.
    Bar (sklearn:_gpr.py:574-652 via gpry/gpr.py:876-881): a theta's value and gradient are the SAME BITS whatever shares the
    call with it -- alone (B = 1), in a batch of three, in the full batch, with one / two / three stream groups, with other column
    blocks, from scratch sets full of NaNs --, they agree with the latency schedule to rounding and with the oracle within
    the tolerances of the other objective tests, and a not-positive-definite theta is reported on its own."""
    from gpry_amd import _lib
    rng, X, y, alpha = _problem(N, d, 11 * N + d + kid)
    base = np.log(np.array([2.0] + [0.5] * d))
    thetas = base + rng.uniform(-0.7, 0.7, (B, d + 1))
    bad = 2
    for j in range(0, 12, 2):
        X[j + 1] = X[j]
    alpha[:12] = 1e-2
    thetas[bad, 0] = np.log(1e15)
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, alpha)
        dv.set_theta(kid, base)
        lat = dv.lml_batch(thetas, True)
        dv.set_option("lml_schedule", 1)
        lml, grad, info = dv.lml_batch(thetas, True)
        assert info[bad] > 0 and np.isneginf(lml[bad]) and not grad[bad].any()
        good = [b for b in range(B) if b != bad]
        assert all(info[b] == 0 for b in good)
        # against the latency schedule (another summation order) and the oracle
        for b in good:
            assert abs(lml[b] - lat[0][b]) <= 1e-11 * max(1.0, abs(lat[0][b])), (b, lml[b], lat[0][b])
            assert np.max(np.abs(grad[b] - lat[1][b])) <= 1e-8 * max(1.0, np.max(np.abs(lat[1][b])))
        for b in ([0, 1] if N <= 1100 else [0]):
            rl, rg = (orc.log_marginal_likelihood(X, y, alpha, thetas[b], kid, eval_gradient=True) if N <= 1100 else
                      orc.log_marginal_likelihood_blocked(X, y, alpha, thetas[b], kid))
            assert abs(lml[b] - rl) <= 1e-10 * max(1.0, abs(rl)), (lml[b], rl)
            assert np.max(np.abs(grad[b] - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg))), (grad[b], rg)

        def same(res, idx):
            for k, b in enumerate(idx):
                assert res[2][k] == info[b]
                assert res[0][k] == lml[b] or (np.isneginf(res[0][k]) and np.isneginf(lml[b])), (b, res[0][k], lml[b])
                np.testing.assert_array_equal(res[1][k], grad[b])

        for b in (0, bad, B - 1):                       # alone
            same(dv.lml_batch(thetas[b:b + 1], True), [b])
        same(dv.lml_batch(thetas[1:4], True), [1, 2, 3])   # three of them, the failing one in the middle
        for streams in (1, 3):                          # other stream groups
            dv.set_option("lml_streams", streams)
            same(dv.lml_batch(thetas, True), list(range(B)))
        dv.set_option("lml_streams", 2)
        dv.set_option("tp_block", 256); dv.set_option("tp_tail", 256)      # other column blocks: the factor is the same, bit for bit
        same(dv.lml_batch(thetas, True), list(range(B)))
        dv.set_option("tp_block", 512); dv.set_option("tp_tail", 1024)
        dv.set_option("panel_debug", 128)               # scratch sets full of NaNs
        same(dv.lml_batch(thetas, True), list(range(B)))
        dv.set_option("panel_debug", 0)
        lml0, info0 = dv.lml_batch(thetas, False)       # value only
        for b in range(B):
            assert info0[b] == info[b] and (lml0[b] == lml[b] or (np.isneginf(lml0[b]) and np.isneginf(lml[b])))
        # the latency schedule is untouched by all this
        dv.set_option("lml_schedule", 0)
        again = dv.lml_batch(thetas, True)
        np.testing.assert_array_equal(again[0], lat[0]); np.testing.assert_array_equal(again[1], lat[1])
    finally:
        dv.close()


def test_throughput_schedule_at_the_headline_size():
    """N = 4096, d = 16 (BASELINE configs[2]): column blocks of 512 with the SYRK engine behind each, whole-tile V = L^-1 and
    K^-1 = V^T V; B-invariant, to rounding the latency schedule, the blocked oracle's value and gradient."""
    from gpry_amd import _lib
    rng, X, y, alpha = _problem(4096, 16, 4112)
    base = np.log(np.array([2.0] + [0.5] * 16))
    thetas = base + rng.uniform(-0.4, 0.4, (5, 17))
    dv = _lib.Device(0)
    try:
        dv.set_train(X, y, alpha); dv.set_theta(3, base)
        lat = dv.lml_batch(thetas, True)
        dv.set_option("lml_schedule", 1)
        lml, grad, info = dv.lml_batch(thetas, True)
        assert not info.any()
        one = dv.lml_batch(thetas[3:4], True)
        assert one[0][0] == lml[3]
        np.testing.assert_array_equal(one[1][0], grad[3])
        assert np.max(np.abs(lml - lat[0]) / np.abs(lat[0])) <= 1e-11
        assert np.max(np.abs(grad - lat[1]).max(1) / np.abs(lat[1]).max(1)) <= 1e-8
        rl, rg = orc.log_marginal_likelihood_blocked(X, y, alpha, thetas[0], 3)
        assert abs(lml[0] - rl) <= 1e-10 * max(1.0, abs(rl))
        assert np.max(np.abs(grad[0] - rg)) <= 1e-7 * max(1.0, np.max(np.abs(rg)))
    finally:
        dv.close()
