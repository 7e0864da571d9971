import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from gpry_amd import _lib
dev = _lib.Device(0)
for N in (6144, 7168, 8192):
    d = 20
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    for mx, caps in ((5120, 0x21), (8192, 0x21), (8192, 0x32), (8192, 0x43)):
        dev.set_option("chol_overlap_max", mx); dev.set_option("chol_caps", caps)
        assert dev.factorize() == 0
        dev.timing_reset()
        for _ in range(4):
            assert dev.factorize() == 0
        print(f"N={N} overlap_max {mx} caps {caps:#x}: potrf {dev.timing('potrf')[0] / 4 * 1e3:8.1f} us", flush=True)
        dev.set_option("timing", 0)
