import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
dev = _lib.Device(0)
for N, d in ((4096, 16), (8192, 20)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4))
    theta = np.log(np.array([4.0] + [0.3] * d)); dev.set_theta(3, theta)
    for lds in (0, 32768, 0, 32768):
        dev.set_option("syrk_lds", lds)
        dev.factorize(); dev.timing_reset()
        for _ in range(6): dev.factorize()
        print(N, "syrk_lds", lds, "potrf %.3f ms" % (dev.timing("potrf")[0] / 6), flush=True)
