import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
dev = _lib.Device(0)
dev.set_option("timing", 0)
N, d = 100, 3
rng = np.random.default_rng(N)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
dev.set_train(X, y, np.full(N, 1e-4)); theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
for _ in range(5): dev.lml(theta, True)
for _ in range(3): dev.factorize()
