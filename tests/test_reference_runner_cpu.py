"""Drop-in check against the REAL reference driver: ``gpry.Runner`` (imported from
/root/reference, build container only) runs its whole active-learning loop -- initial training,
NORA acquisition, SVM classifier, hyper-parameter fits, convergence criteria -- on top of the
``gpry_amd`` classes after ``gpry_amd.integration.patch_gpry()``, with the oracle-backed device
double standing in for the GPU.  Skipped where the reference is not mounted (the GPU box)."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(runner_cls, loglike, bounds, seed, max_total=120):
    r = runner_cls(loglike, bounds, gpr={"kernel": {"Matern": {"nu": 2.5}}},
                   gp_acquisition={"NORA": {"sampler": "uniform", "mc_every": 2}},
                   options={"max_total": max_total, "max_finite": max_total}, checkpoint=None, verbose=0,
                   seed=seed)
    # the final MC sample + diagnosis need samplers (Cobaya / nested samplers) that are not
    # installed here and are outside the hot path: stop the driver right after convergence
    r.generate_mc_sample = lambda *a, **k: None
    r.diagnose_last_mc_sample = lambda *a, **k: True
    r.run()
    return r


def test_reference_runner_drives_the_mirror_classes(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_goldens import import_reference
    import_reference()
    import scipy.stats as st
    import gpry.run
    rv = st.multivariate_normal([0.5, -0.3], [[1.0, 0.6], [0.6, 0.8]])

    def loglike(x, y):
        return rv.logpdf([x, y])

    bounds = [[-5, 5], [-5, 5]]
    ref = _run(gpry.run.Runner, loglike, bounds, seed=1)          # the reference on its own classes
    assert ref.has_converged

    from oracle_device import OracleDevice
    from gpry_amd import _lib
    import gpry_amd.gpr as mirror_gpr
    import gpry_amd.gp_acquisition as mirror_acq
    saved = {k: getattr(gpry.run, k) for k in ("GaussianProcessRegressor", "GenericGPAcquisition",
                                                "Normalize_bounds", "Normalize_y")}
    saved_nora, saved_gpr = gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor
    saved_bo = gpry.gp_acquisition.BatchOptimizer
    monkeypatch.setattr(_lib, "Device", lambda index=0: OracleDevice())
    try:
        from gpry_amd.integration import patch_gpry
        patch_gpry()
        ours = _run(gpry.run.Runner, loglike, bounds, seed=1)
    finally:
        for k, v in saved.items():
            setattr(gpry.run, k, v)
        gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor = saved_nora, saved_gpr
        gpry.gp_acquisition.BatchOptimizer = saved_bo
    assert isinstance(ours.gpr, mirror_gpr.GaussianProcessRegressor)
    assert isinstance(ours.acquisition, mirror_acq.NORA)
    assert isinstance(ours.gpr.device, OracleDevice)
    assert ours.has_converged
    # same driver, same seeds: the two runs need a similar number of truth evaluations and end
    # with surrogates that agree with the truth (and with each other) around the mode
    assert abs(ours.gpr.n_total - ref.gpr.n_total) <= 8
    Xt = rv.rvs(50, random_state=3)
    truth = np.array([loglike(*x) for x in Xt]) - np.log(100.0)   # log-posterior: uniform prior on the box
    err_ref = np.max(np.abs(ref.gpr.predict(Xt) - truth))
    err_ours = np.max(np.abs(ours.gpr.predict(Xt) - truth))
    assert err_ref < 0.05 and err_ours < 0.05
    np.testing.assert_allclose(ours.mean, ref.mean, atol=0.15)
    # the classifier (account_for_inf="SVM" is the Runner's default) and the counters were exercised
    assert ours.gpr.infinities_classifier is not None and ours.gpr.n_eval > 0


def test_reference_runner_with_an_infinite_region_uses_the_device_gates(monkeypatch):
    """3-d posterior with a half-space of -inf: the Runner's default SVM classifier is trained by the
    mirror's wrapper and its verdicts for the NORA pools come from the device gates."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_goldens import import_reference
    import_reference()
    import scipy.stats as st
    import gpry.run
    rv = st.multivariate_normal(np.zeros(3), np.diag([0.5, 1.0, 0.8]))

    def loglike(a, b, c):
        return -np.inf if a + b > 2.5 else rv.logpdf([a, b, c])

    bounds = [[-6, 6]] * 3
    ref = _run(gpry.run.Runner, loglike, bounds, seed=2, max_total=300)
    from oracle_device import OracleDevice
    from gpry_amd import _lib
    saved = {k: getattr(gpry.run, k) for k in ("GaussianProcessRegressor", "GenericGPAcquisition",
                                                "Normalize_bounds", "Normalize_y")}
    saved_nora, saved_gpr = gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor
    saved_bo = gpry.gp_acquisition.BatchOptimizer
    monkeypatch.setattr(_lib, "Device", lambda index=0: OracleDevice())
    try:
        from gpry_amd.integration import patch_gpry
        patch_gpry()
        ours = _run(gpry.run.Runner, loglike, bounds, seed=2, max_total=300)
    finally:
        for k, v in saved.items():
            setattr(gpry.run, k, v)
        gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor = saved_nora, saved_gpr
        gpry.gp_acquisition.BatchOptimizer = saved_bo
    assert ref.has_converged and ours.has_converged
    assert ours.gpr.n_total > ours.gpr.n                      # some evaluations hit the -inf region
    assert ours.gpr.device.gates is not None                  # ... and the sweep used the device gates
    assert abs(ours.gpr.n_total - ref.gpr.n_total) <= 15
    Xt = rv.rvs(60, random_state=4)
    Xt = Xt[Xt[:, 0] + Xt[:, 1] < 2.0]
    truth = np.array([loglike(*x) for x in Xt]) - 3 * np.log(12.0)
    assert np.max(np.abs(ours.gpr.predict(Xt) - truth)) < 0.1
    assert np.max(np.abs(ref.gpr.predict(Xt) - truth)) < 0.1


@pytest.mark.parametrize("which", ["reference", "mirror"])
def test_reference_batch_optimizer_runs_on_the_mirror_gpr(monkeypatch, which):
    """The reference's own gradient-based acquisition (``BatchOptimizer``, gp_acquisition.py:270-389)
    needs nothing but ``predict(..., return_mean_grad, return_std_grad)``, lies appended with
    ``fit_gpr=False`` and deep copies: on top of the mirror GPR it drives the run to convergence
    (``which="reference"``).  ``patch_gpry()`` also rebinds the name ``BatchOptimizer`` that
    ``Runner`` looks up (run.py:392): ``which="mirror"`` is the same run on the mirror class."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_goldens import import_reference
    import_reference()
    import scipy.stats as st
    import gpry.run
    import gpry.gp_acquisition
    rv = st.multivariate_normal([0.5, -0.3], [[1.0, 0.6], [0.6, 0.8]])

    def loglike(x, y):
        return rv.logpdf([x, y])

    from oracle_device import OracleDevice
    from gpry_amd import _lib
    import gpry_amd.gpr as mirror_gpr
    saved = {k: getattr(gpry.run, k) for k in ("GaussianProcessRegressor", "GenericGPAcquisition",
                                                "Normalize_bounds", "Normalize_y")}
    saved_nora, saved_gpr = gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor
    saved_bo = gpry.gp_acquisition.BatchOptimizer
    monkeypatch.setattr(_lib, "Device", lambda index=0: OracleDevice())
    try:
        from gpry_amd.integration import patch_gpry
        patch_gpry()
        import gpry_amd.gp_acquisition as mirror_acq
        assert gpry.gp_acquisition.BatchOptimizer is mirror_acq.BatchOptimizer
        if which == "reference":
            gpry.gp_acquisition.BatchOptimizer = saved_bo
        r = gpry.run.Runner(loglike, [[-5, 5], [-5, 5]], gpr={"kernel": {"Matern": {"nu": 2.5}}},
                            gp_acquisition="BatchOptimizer", options={"max_total": 120, "max_finite": 120},
                            checkpoint=None, verbose=0, seed=1)
        r.generate_mc_sample = lambda *a, **k: None
        r.diagnose_last_mc_sample = lambda *a, **k: True
        r.run()
    finally:
        for k, v in saved.items():
            setattr(gpry.run, k, v)
        gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor = saved_nora, saved_gpr
        gpry.gp_acquisition.BatchOptimizer = saved_bo
    assert isinstance(r.gpr, mirror_gpr.GaussianProcessRegressor)
    assert type(r.acquisition).__name__ == "BatchOptimizer" and r.has_converged
    assert (type(r.acquisition) is mirror_acq.BatchOptimizer) == (which == "mirror")
    Xt = rv.rvs(30, random_state=3)
    truth = np.array([loglike(*x) for x in Xt]) - np.log(100.0)
    assert np.max(np.abs(r.gpr.predict(Xt) - truth)) < 0.1


def test_reference_checkpoint_is_light_and_resumes_on_the_mirror_classes(monkeypatch, tmp_path):
    """SURVEY.md 8(f)4, second half: the reference's own checkpoint (``gpry/io.py:110-151``, dill
    pickles written by ``Runner.save_checkpoint``, ``run.py:736``) of a mirror GPR holds the training
    set and theta only -- no ``L_`` / ``V_`` / device handles -- and ``load_checkpoint="resume"``
    (``run.py:231-262``) rebuilds the device state and carries the run on to convergence."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_goldens import import_reference
    import_reference()
    import dill
    import scipy.stats as st
    import gpry.run
    import gpry.gp_acquisition
    rv = st.multivariate_normal([0.5, -0.3], [[1.0, 0.6], [0.6, 0.8]])

    def loglike(x, y):
        return rv.logpdf([x, y])

    from oracle_device import OracleDevice
    from gpry_amd import _lib
    import gpry_amd.gpr as mirror_gpr
    saved = {k: getattr(gpry.run, k) for k in ("GaussianProcessRegressor", "GenericGPAcquisition",
                                                "Normalize_bounds", "Normalize_y")}
    saved_nora, saved_gpr = gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor
    saved_bo = gpry.gp_acquisition.BatchOptimizer
    monkeypatch.setattr(_lib, "Device", lambda index=0: OracleDevice())

    def make(load, max_total):
        r = gpry.run.Runner(loglike, [[-5, 5], [-5, 5]], gpr={"kernel": {"Matern": {"nu": 2.5}}},
                            gp_acquisition={"NORA": {"sampler": "uniform", "mc_every": 2}},
                            options={"max_total": max_total, "max_finite": max_total, "n_initial": 6,
                                     "max_initial": 12},
                            checkpoint=str(tmp_path), load_checkpoint=load, verbose=0, seed=1)
        r.generate_mc_sample = lambda *a, **k: None
        r.diagnose_last_mc_sample = lambda *a, **k: True
        return r

    try:
        from gpry_amd.integration import patch_gpry
        patch_gpry()
        first = make("overwrite", 12)           # budget exhausted after the initial set: not converged
        first.run()
        assert not first.has_converged and first.gpr.n_total == 12
        size = os.path.getsize(tmp_path / "gpr.pkl")
        assert size < 20_000, size               # training set + theta + bookkeeping only
        with open(tmp_path / "gpr.pkl", "rb") as f:
            state = dill.load(f).__dict__
        assert state["_dev"] is None and state["_kb"] is None and not state["_host_factor"]
        assert not state["_dev_factor_ok"] and not {"L_", "V_", "alpha_"} & set(state)
        resumed = make("resume", 12)
        assert resumed.loaded_from_checkpoint
        assert isinstance(resumed.gpr, mirror_gpr.GaussianProcessRegressor) and resumed.gpr.n_total == 12
        np.testing.assert_array_equal(resumed.gpr.X_train, first.gpr.X_train)
        Xt = rv.rvs(20, random_state=5)
        np.testing.assert_allclose(resumed.gpr.predict(Xt), first.gpr.predict(Xt), rtol=1e-12)
        resumed.max_total = resumed.max_finite = 120      # the loaded options carry the old budget
        resumed._mc_options = None                        # the reference does not restore it on resume
        resumed.run()
    finally:
        for k, v in saved.items():
            setattr(gpry.run, k, v)
        gpry.gp_acquisition.NORA, gpry.gpr.GaussianProcessRegressor = saved_nora, saved_gpr
        gpry.gp_acquisition.BatchOptimizer = saved_bo
    assert resumed.has_converged and resumed.gpr.n_total > 12
    truth = np.array([loglike(*x) for x in Xt]) - np.log(100.0)
    assert np.max(np.abs(resumed.gpr.predict(Xt) - truth)) < 0.1
