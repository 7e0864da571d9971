"""Rebind GPry's class names to the device-backed mirrors."""


def patch_gpry():
    """After this call ``gpry.Runner`` builds / accepts the MI355X classes.

    Rebinds the names that ``gpry/run.py`` looks up (``GaussianProcessRegressor`` at
    :308,346; acquisition classes by name at :392 -- ``NORA`` and ``BatchOptimizer``;
    ``GenericGPAcquisition`` at :360; the default pre-processors at :318-319).  Returns
    the patched ``gpry`` module.
    """
    import gpry
    import gpry.gpr
    import gpry.gp_acquisition
    import gpry.run
    from gpry_amd.gp_acquisition import NORA, BatchOptimizer, GenericGPAcquisition
    from gpry_amd.gpr import GaussianProcessRegressor
    from gpry_amd.preprocessing import Normalize_bounds, Normalize_y

    gpry.gpr.GaussianProcessRegressor = GaussianProcessRegressor
    gpry.run.GaussianProcessRegressor = GaussianProcessRegressor
    gpry.gp_acquisition.NORA = NORA
    gpry.gp_acquisition.BatchOptimizer = BatchOptimizer
    gpry.run.GenericGPAcquisition = GenericGPAcquisition
    gpry.run.Normalize_bounds = Normalize_bounds
    gpry.run.Normalize_y = Normalize_y
    return gpry
