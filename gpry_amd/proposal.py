"""Starting points for the optimisation of the acquisition function (gpry/proposal.py).

Only what ``BatchOptimizer`` needs: the proposer interface (``get`` / ``update`` / ``update_bounds``),
the uniform proposer, the centroids proposer and their mixture -- the default of
gpry/gp_acquisition.py:219-221.  A seeded run has to propose the reference's points, so the draws consume the
generator exactly as the reference's frozen scipy distributions do -- ``uniform(loc, scale).rvs`` is
``uniform(0, 1, size) * scale + loc`` and ``expon(scale=s).rvs`` is ``standard_exponential(size) * s`` on the
generator that is passed in -- but are written as those two expressions (a frozen-distribution ``rvs`` costs 50 us
of argument checking per call; tests/test_fit_farm_cpu.py compares both with scipy bit for bit).
"""
import numpy as np

from gpry_amd.tools import check_random_state, is_in_bounds


class Proposer:
    """Interface of gpry/proposal.py:45-88; keeps the box as ``bounds`` (d x 2, float)."""
    bounds = None

    def get(self, rng=None):
        raise NotImplementedError("a proposer returns one point of the box per call")

    def update(self, gpr):
        """Most proposers do not look at the surrogate model."""

    def update_bounds(self, bounds):
        self.bounds = np.array(bounds, dtype=float, copy=True)

    @property
    def d(self):
        return self.bounds.shape[0]

    @property
    def corners(self):
        return self.bounds[:, 0], self.bounds[:, 1]


class UniformProposer(Proposer):
    """Uniform in the box (gpry/proposal.py:136-160)."""

    def __init__(self, bounds):
        self.update_bounds(bounds)

    def get(self, rng=None):
        lo, hi = self.corners
        return check_random_state(rng).uniform(0.0, 1.0, self.d) * (hi - lo) + lo


class CentroidsProposer(Proposer):
    """Centroid of d+1 random training points, pushed in every dimension along the difference to one of them
    by an exponentially distributed factor, clipped to the box (gpry/proposal.py:258-319)."""

    def __init__(self, bounds, lambd=1.0):
        self.kick_scale = 1 / lambd
        self.training = None            # all training locations / those inside the current box
        self.training_ = None
        self.update_bounds(bounds)

    def update(self, gpr):
        self.training = np.array(gpr.X_train, copy=True)

    def update_bounds(self, bounds):
        Proposer.update_bounds(self, bounds)
        if self.training is not None:
            self.training_ = self.training[is_in_bounds(self.training, self.bounds)]

    def _simplex(self, rng):
        """d+1 distinct training points, from inside the box if it holds that many (one draw either way is what
        the reference consumes: a failed ``choice`` raises before it touches the generator)."""
        for pool in (self.training_, self.training):
            try:
                return pool[rng.choice(len(pool), size=self.d + 1, replace=False)]
            except ValueError:
                if pool is self.training:
                    raise
        raise AssertionError("unreachable")

    def get(self, rng=None):
        rng = check_random_state(rng)
        vertices = self._simplex(rng)
        centre = vertices.mean(axis=0)
        toward = vertices[rng.choice(self.d + 1, size=self.d, replace=False), np.arange(self.d)] - centre
        toward *= rng.standard_exponential(self.d) * self.kick_scale
        return np.clip(centre + toward, *self.corners)


class PartialProposer(Proposer):
    """``true_proposer`` with a fraction of uniform draws mixed in (gpry/proposal.py:163-215)."""

    def __init__(self, bounds, true_proposer, random_proposal_fraction=0.25):
        if not 0.0 <= random_proposal_fraction <= 1.0:
            raise ValueError("Cannot pass a fraction outside of [0,1]. You passed "
                             f"'random_proposal_fraction={random_proposal_fraction}'")
        if not isinstance(true_proposer, Proposer):
            raise ValueError("The true proposer needs to be a valid proposer.")
        self.rpf = random_proposal_fraction
        self.random_proposer, self.true_proposer = UniformProposer(bounds), true_proposer

    @property
    def _both(self):
        return self.random_proposer, self.true_proposer

    def get(self, rng=None):
        rng = check_random_state(rng)
        uniform, informed = self._both
        return (informed if rng.random() > self.rpf else uniform).get(rng=rng)

    def update(self, gpr):
        self.true_proposer.update(gpr)

    def update_bounds(self, bounds):
        for p in self._both:
            p.update_bounds(bounds)
