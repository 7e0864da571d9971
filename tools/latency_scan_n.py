import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib
for N, d in ((64, 2), (128, 4), (256, 4), (384, 4), (512, 4), (1024, 8)):
    rng = np.random.default_rng(0)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    dev = _lib.Device(0)
    dev.set_train(X, y, np.full(N, 1e-4))
    th = np.log(np.array([4.0] + [0.3] * d))
    dev.set_theta(3, th)
    assert dev.factorize() == 0
    def t(fn, reps=50):
        for _ in range(3): fn()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps * 1e6
    x1 = rng.uniform(0, 1, (1, d))
    a = t(lambda: dev.lml(th, True)); b = t(lambda: dev.predict(x1)); c = t(lambda: dev.predict(x1, return_std=True))
    dev.timing_reset()
    for _ in range(10): dev.lml(th, True)
    st = {k: round(dev.timing(k)[0] / 10 * 1e3, 1) for k in ("kernel_build", "potrf", "trtri", "lauum", "lml_traces")}
    print(f"N={N}: lml+grad {a:.0f} us, predict1 {b:.0f} us, predict1+std {c:.0f} us; stages {st}", flush=True)
    dev.close()
