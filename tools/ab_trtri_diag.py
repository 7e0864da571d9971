import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
dev = _lib.Device(0)
for N, d in ((100, 3), (1024, 8), (4096, 16)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
    res = {}
    for v1 in (1, 2):
        dev.set_option("trtri_diag_v1", v1)
        assert dev.factorize() == 0
        L, V, a = dev.get_factor()
        dev.timing_reset()
        for _ in range(10): dev.factorize()
        res[v1] = (V, dev.timing("trtri")[0] / 10)
        print(N, "v1" if v1 else "v2", "max|VL-I| = %.2e" % np.max(np.abs(V @ L - np.eye(N))), "trtri %.3f ms" % res[v1][1], flush=True)
    print("   max |V_v2 - V_v1| / max|V| = %.2e" % (np.max(np.abs(res[2][0] - res[1][0])) / np.max(np.abs(res[1][0]))))
