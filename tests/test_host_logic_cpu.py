"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the
product refuses to run without a device, and the host-side logic (kernel parameter
algebra, pre-processors, LogExp, RankedPool control flow, shortlist merge across ranks)
is checked with the oracle standing in for the device."""
import os
import re
from functools import partial

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import gpry_oracle as orc


def test_library_exports_every_declared_symbol():
    from gpry_amd import _lib
    header = open(os.path.join(ROOT, "include", "gpry_hip.h")).read()
    declared = set(re.findall(r"\b(gpry_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = _lib.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/gpry_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes table and header disagree"
    assert lib.gpry_version() >= 100


def test_every_option_of_the_library_is_documented_in_the_header():
    """ONE option table (csrc/ctx.hip) behind gpry_ctx_set_option / gpry_ctx_get_option; include/gpry_hip.h documents every key
    (a maintainer binds against the header alone), and the options a fit copies to its extra contexts exist."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keys = re.findall(r'OPT_INT\("([a-z0-9_]+)"', open(os.path.join(root, "gpry_amd", "csrc", "ctx.hip")).read())
    header = open(os.path.join(root, "include", "gpry_hip.h")).read()
    assert len(keys) == len(set(keys)) and len(keys) >= 30
    missing = [k for k in keys if f'"{k}"' not in header]
    assert not missing, f"options without a line in include/gpry_hip.h: {missing}"
    from gpry_amd import gpr as G
    assert set(G._FIT_CONTEXT_OPTIONS) <= set(keys)


def test_product_fails_loudly_without_gpu():
    from gpry_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.GpryHipError, match="no CPU fallback"):
        _lib.Device(0)
    from gpry_amd.gpr import GaussianProcessRegressor
    b = np.array([[0.0, 1.0]] * 2)
    gpr = GaussianProcessRegressor(kernel="RBF", bounds=b, account_for_inf=None)
    with pytest.raises(_lib.GpryHipError):
        gpr.append_to_data(np.random.rand(5, 2), np.random.rand(5), fit_gpr=False)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gpry_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("test oracle", ""), f"{fn} mentions the oracle"


@pytest.mark.parametrize("kid,name", [(0, "RBF"), (3, {"Matern": {"nu": 2.5}})])
def test_kernel_parameter_algebra(kid, name):
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y
    g = load_golden("fit")
    p = f"f6_k{kid}_"
    b = g[p + "bounds"]
    gpr = GaussianProcessRegressor(kernel=name, bounds=b, preprocessing_X=Normalize_bounds(b),
                                   preprocessing_y=Normalize_y(), account_for_inf=None)
    k = gpr.kernel
    np.testing.assert_allclose(k.bounds, g[p + "theta_bounds"], rtol=1e-15)
    d = len(b)
    np.testing.assert_allclose(k.theta, np.log([10.0] + [0.1] * d), rtol=1e-14)
    th = np.log(np.array([3.0] + [0.2 + 0.1 * i for i in range(d)]))
    k2 = k.clone_with_theta(th)
    np.testing.assert_allclose(k2.theta, th)
    np.testing.assert_allclose(k.theta, np.log([10.0] + [0.1] * d))   # original untouched
    assert k2.k1.constant_value == pytest.approx(3.0)
    np.testing.assert_allclose(k2.k2.length_scale, np.exp(th[1:]))
    kid_dev, th_full = k2.device_spec(d)
    assert kid_dev == kid
    np.testing.assert_allclose(th_full, th)
    np.testing.assert_allclose(k2.grad_from_full(np.arange(d + 1.0), d), np.arange(d + 1.0))
    np.testing.assert_allclose(k2.diag(np.zeros((4, d))), 3.0)
    assert k2.n_dims == d + 1 and len(k2.hyperparameters) == 2


def test_isotropic_and_fixed_hyperparameters_map_to_device_theta():
    from gpry_amd.kernels import ConstantKernel, RBF
    k = ConstantKernel(2.0, "fixed") * RBF(0.5, (1e-3, 10.0))
    assert k.n_dims == 1
    kid, full = k.device_spec(3)
    np.testing.assert_allclose(full, np.log([2.0, 0.5, 0.5, 0.5]))
    np.testing.assert_allclose(k.grad_from_full(np.array([9.0, 1.0, 2.0, 3.0]), 3), [6.0])
    np.testing.assert_allclose(k.theta_to_full(np.log([0.25]), 3), np.log([2.0, 0.25, 0.25, 0.25]))


def test_preprocessors_and_tools():
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y, DummyPreprocessor
    from gpry_amd.tools import get_Xnumber, is_in_bounds, shrink_bounds
    b = np.array([[-5.0, 5.0], [0.0, 2.0]])
    nb = Normalize_bounds(b)
    X = np.array([[0.0, 1.0], [-5.0, 2.0]])
    np.testing.assert_allclose(nb.transform(X), [[0.5, 0.5], [0.0, 1.0]])
    np.testing.assert_allclose(nb.inverse_transform(nb.transform(X)), X)
    np.testing.assert_allclose(nb.transform_bounds(b), [[0, 1], [0, 1]])
    ny = Normalize_y()
    with pytest.raises(TypeError):
        ny.transform(np.ones(3))
    y = np.array([1.0, 2.0, 4.0, -np.inf])
    ny.fit(None, y)
    assert ny.mean_ == pytest.approx(7 / 3) and ny.std_ == pytest.approx(np.std([1, 2, 4]))
    np.testing.assert_allclose(ny.inverse_transform(ny.transform(y[:3])), y[:3])
    assert DummyPreprocessor.transform(3) == 3
    assert get_Xnumber("5d", "d", 4, int) == 20 and get_Xnumber("30d1.5", "d", 4, int) == 240
    assert get_Xnumber("20s", "s", None, dtype=float) == (20.0, True, None)
    assert list(is_in_bounds(X, b)) == [True, True] and not is_in_bounds([[6, 1]], b)[0]
    got = shrink_bounds(b, np.array([[0.0, 0.5], [2.0, 1.5]]), factor=2)
    np.testing.assert_allclose(got, orc.shrink_bounds(b, np.array([[0.0, 0.5], [2.0, 1.5]]), 2))


def test_logexp_value_path_vs_reference_vectors():
    from gpry_amd.acquisition_functions import LogExp
    g = load_golden("predict")
    af = LogExp(dimension=7)
    assert af.zeta == float(g["f5_zeta"])
    acq = LogExp.f(g["f5_mu"], g["f5_std"], float(g["f5_baseline"]), float(g["f5_noise"]), af.zeta)
    assert np.array_equal(np.isneginf(acq), np.isneginf(g["f5_acq"]))
    fin = np.isfinite(acq)
    np.testing.assert_array_equal(acq[fin], g["f5_acq"][fin])


# ---- RankedPool / NORA control flow with the oracle standing in for the device --------------
class FakeDevice:
    """Test double of gpry_amd._lib.Device for the sweep/top-k calls (numpy via the oracle)."""

    def __init__(self, model):
        self.m = model

    def sweep_logexp(self, X, zeta, baseline, sigma_n, mask=None, M=None, want=()):
        X = self._lastX if X is None else X     # X=None: candidate set still resident
        self._lastX = X
        y, s =self.m.predict(X, return_std=True) if len(X) else (np.empty(0), np.empty(0))
        self.acq = orc.logexp_f(y, s, baseline, sigma_n, zeta)
        self.y, self.s = y, s
        return {"y": y, "sigma": s, "acq": self.acq, "n_nan": int(np.isnan(self.acq).sum())}

    def sweep_topk(self, K, exclude=None):
        from gpry_amd._lib import CAND_DTYPE
        M = len(self.acq)
        ok = np.ones(M, bool)
        if exclude is not None:
            ok[np.asarray(exclude, dtype=int)] = False
        order = np.lexsort((-np.arange(M), -self.acq))
        order = order[ok[order]]
        top = np.zeros(min(K, len(order)), dtype=CAND_DTYPE)
        sel = order[:K]
        top["acq"], top["y"], top["sigma"], top["idx"] = self.acq[sel], self.y[sel], self.s[sel], sel
        bound = self.acq[order[K]] if len(order) > K else -np.inf
        return top, bound


class FakeGroup:
    """Test double of gpry_amd._lib.DeviceGroup: k FakeDevices on contiguous shards, shortlists merged
    with the hold-back rule (a Python restatement of gpry_group_sweep_topk)."""

    def __init__(self, model, k):
        self.members = [FakeDevice(model) for _ in range(k)]
        self.size, self.sweep_epoch, self.n_set_model = k, 0, 0

    def set_model(self, *a):
        self.n_set_model += 1
        return 0

    def set_gates(self, *a, **kw):
        pass

    def sweep_logexp(self, X, zeta, baseline, sigma_n, mask=None, M=None, want=()):
        if X is not None:
            M = len(X)
            per = -(-M // self.size)
            self.cuts = [(min(i * per, M), min((i + 1) * per, M)) for i in range(self.size)]
        outs = [m.sweep_logexp(None if X is None else X[lo:hi], zeta, baseline, sigma_n)
                for m, (lo, hi) in zip(self.members, self.cuts)]
        self.sweep_epoch += 1
        out = {k: np.concatenate([o[k] for o in outs]) for k in ("y", "sigma", "acq")}
        out["n_nan"] = sum(o["n_nan"] for o in outs)
        self._last = out
        return out

    def sweep_fetch(self, want=("y", "sigma")):
        return self._last

    def sweep_topk(self, K, exclude=None):
        parts, gbound, exhausted = [], -np.inf, True
        for m, (lo, hi) in zip(self.members, self.cuts):
            if hi <= lo:
                continue
            ex = None if exclude is None else np.asarray([e - lo for e in exclude if lo <= e < hi], dtype=int)
            top, bound = m.sweep_topk(K, exclude=ex)
            top = top.copy()
            top["idx"] += lo
            parts.append(top)
            gbound = max(gbound, bound)
            exhausted = exhausted and len(top) < K
        merged = np.concatenate(parts)
        merged = merged[np.lexsort((-merged["idx"], -merged["acq"]))]
        if not exhausted:
            merged = merged[merged["acq"] > gbound]
        return merged, (-np.inf if exhausted else gbound), exhausted


class FakeGPR:
    """Quacks like gpry_amd.gpr.GaussianProcessRegressor where NORA / RankedPool touch it."""

    def __init__(self, model):
        self.m = model
        self.device = FakeDevice(model)
        self.d, self.n_eval, self.noise_level = model.d, 0, model.noise_level
        self.infinities_classifier = None
        self._factor_epoch = 1
        self.X_train_ = self.y_train_ = self.alpha = None      # what a device group would replicate

    def _device_theta(self):
        return 0, None

    def _affine_args(self):
        return None

    y_max = property(lambda self: self.m.y_max)
    n = property(lambda self: self.m.n)

    def _ensure_factor(self):
        pass

    _push_affine = _ensure_factor

    def _masks(self, X, validate, ignore):
        return None

    def predict(self, X, return_std=False, validate=True):
        return self.m.predict(X, return_std=return_std)

    def predict_std(self, X, validate=True):
        return self.m.predict_std(X)

    def conditioned(self, X, y):
        c = FakeGPR(self.m.conditioned_copy(X, y))
        return c

    def append_to_data(self, X, y, **kw):
        self.m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
        self._factor_epoch += 1


def _golden_model(tag):
    g = load_golden("multi_add")
    p = f"f7{tag}_"
    N, d = g[p + "X"].shape
    bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, int(g[p + "M"]), int(g[p + "seed"]))
    m = orc.OracleGPR(bounds, kernel_id=int(g[p + "kid"]))
    m.theta = np.array(g[p + "theta"])
    m.fitted = True
    m.append_to_data(X, y, fit_gpr=False, fit_preprocessors=True)
    return g, p, bounds, Xc, m


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("shortlist", [4, 64, 100000])
def test_nora_shortlist_stream_reproduces_reference_pool(tag, shortlist):
    """The streamed shortlist (any initial size, incl. too small ones that must be
    extended) gives exactly the reference's proposals, twice in a row."""
    from gpry_amd.gp_acquisition import NORA
    g, p, bounds, Xc, m = _golden_model(tag)
    gpr = FakeGPR(m)
    npts = len(g[p + "acq_cond"]) - 1
    acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, shortlist_size=shortlist)
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    np.testing.assert_array_equal(Xp, g[p + "X_pool"])
    np.testing.assert_allclose(yp, g[p + "y_pool"], rtol=1e-9)
    np.testing.assert_allclose(ap, g[p + "acq_pool"], rtol=1e-8)
    np.testing.assert_allclose(acq.pool.acq_cond, g[p + "acq_cond"], rtol=1e-6)
    assert acq.pool.cache_counter == int(g[p + "cache_counter"])
    gpr.append_to_data(Xp, g[p + "y_new"])
    Xp2, yp2, ap2 = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    np.testing.assert_array_equal(Xp2, g[p + "X_pool2"])
    np.testing.assert_allclose(ap2, g[p + "acq_pool2"], rtol=1e-7)
    assert len(acq.last_MC_sample(warn_reweight=False)[1]) == int(g[p + "n_rw"])


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("k,shortlist", [(2, 4), (3, 64), (8, 4)])
def test_nora_over_a_device_group_reproduces_reference_pool(tag, k, shortlist):
    """``NORA(devices=<group>)``: the protocol between multi_add and a device group (global exclusion
    rows, shortlist extension with held-back entries, resident pool, reweighted second call)."""
    from gpry_amd.gp_acquisition import NORA
    g, p, bounds, Xc, m = _golden_model(tag)
    gpr = FakeGPR(m)
    grp = FakeGroup(m, k)
    npts = len(g[p + "acq_cond"]) - 1
    acq = NORA(bounds, sampler="uniform", mc_every=2, verbose=0, shortlist_size=shortlist, devices=grp)
    acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
    Xp, yp, ap = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    np.testing.assert_array_equal(Xp, g[p + "X_pool"])
    np.testing.assert_allclose(ap, g[p + "acq_pool"], rtol=1e-8)
    np.testing.assert_allclose(acq.pool.acq_cond, g[p + "acq_cond"], rtol=1e-6)
    assert acq.stats["sweep_contexts"] == k and grp.n_set_model == 1
    gpr.append_to_data(Xp, g[p + "y_new"])
    Xp2, yp2, ap2 = acq.multi_add(gpr, n_points=npts, rng=np.random.default_rng(2))
    np.testing.assert_array_equal(Xp2, g[p + "X_pool2"])
    np.testing.assert_allclose(ap2, g[p + "acq_pool2"], rtol=1e-7)
    assert len(acq.last_MC_sample(warn_reweight=False)[1]) == int(g[p + "n_rw"])
    assert grp.n_set_model == 2                      # the model changed once between the calls
    with pytest.raises(ValueError):
        class C:
            world, rank = 2, 0
        NORA(bounds, sampler="uniform", comm=C(), devices=[0, 1])


def test_nora_device_spec_resolution(monkeypatch):
    from gpry_amd import _lib
    from gpry_amd.gp_acquisition import NORA
    b = np.array([[0.0, 1.0]] * 2)
    monkeypatch.setattr(_lib, "device_count", lambda: 4)
    monkeypatch.delenv("GPRY_HIP_DEVICES", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert NORA(b, verbose=0)._resolve_devices() == [0, 1, 2, 3]          # one process: every GPU
    assert NORA(b, verbose=0, devices=2)._resolve_devices() == [0, 1]
    assert NORA(b, verbose=0, devices=[0, 0, 0])._resolve_devices() == [0, 0, 0]
    assert NORA(b, verbose=0, devices=[2])._resolve_devices() is None
    monkeypatch.setenv("WORLD_SIZE", "8")                                   # one process per GPU
    assert NORA(b, verbose=0)._resolve_devices() is None
    monkeypatch.setenv("GPRY_HIP_DEVICES", "1,3")
    assert NORA(b, verbose=0)._resolve_devices() == [1, 3]
    monkeypatch.setenv("GPRY_HIP_DEVICES", "none")
    assert NORA(b, verbose=0)._resolve_devices() is None
    monkeypatch.setattr(_lib, "device_count", lambda: 1)
    monkeypatch.setenv("GPRY_HIP_DEVICES", "all")
    assert NORA(b, verbose=0)._resolve_devices() is None


def test_ranked_pool_mirror_equals_oracle_restatement():
    from gpry_amd.gp_acquisition import RankedPool
    g, p, bounds, Xc, m = _golden_model("a")
    gpr = FakeGPR(m)
    y, s = m.predict(Xc, return_std=True)
    f = partial(orc.logexp_f, baseline=m.y_max, noise_level=m.noise_level, zeta=orc.auto_zeta(m.d))
    a = f(y, s)
    for method in ("single sort acq", "single sort y", "single", "bulk"):
        sub = slice(0, 600)
        mine = RankedPool(3, gpr=gpr, acq_func=f, verbose=0)
        mine.add(Xc[sub], y[sub], s[sub], a[sub], method=method)
        ref = orc.OracleRankedPool(3, m, f)
        ref.add(Xc[sub], y[sub], s[sub], a[sub], method=method)
        np.testing.assert_array_equal(mine.X, ref.X)
        np.testing.assert_allclose(mine.acq_cond, ref.acq_cond, rtol=1e-12)
    c = mine.copy(drop_empty=True)
    assert len(c.X) <= 4 and not hasattr(c, "_gpr")
    with pytest.raises(ValueError):
        mine.add(Xc[:2], y[:2], s[:2], np.array([np.nan, 1.0]))


def test_nora_argument_checks():
    from gpry_amd.gp_acquisition import NORA, NestedSamplerNotInstalledError, builtin_names
    b = np.array([[0.0, 1.0]] * 3)
    with pytest.raises(NestedSamplerNotInstalledError):
        NORA(b, sampler="polychord")
    acq = NORA(b, sampler="uniform", verbose=0)
    assert acq.mc_every == 3 and acq.nlive_max == 75 and acq.num_repeats == 15
    assert acq.acq_func.zeta == 3 ** -0.85
    with pytest.raises(ValueError):
        acq.multi_add(None, n_points=0)
    assert "NORA" in builtin_names()


def test_header_is_plain_c99_and_the_c_client_compiles():
    """include/gpry_hip.h must be consumable by a C compiler without C++ or torch headers."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.c")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                    "-I", os.path.join(ROOT, "include"), src], check=True)
