import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N, d, M = 4096, 16, 262144
rng = np.random.default_rng(0)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N); Xc = rng.uniform(0, 1, (M, d))
dev = _lib.Device(0)
dev.set_train(X, y, np.full(N, 1e-4)); dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
assert dev.factorize() == 0
ref = None
for alt, ps in ((1, 0), (1, 1), (0, 0), (0, 1), (1, 0), (1, 1), (1, 0), (1, 1)):
    dev.set_option("sweep_altwalk", alt); dev.set_option("sweep_persist", ps)
    out = dev.sweep_logexp(Xc, 0.1, 0.0, 1e-2, want=("sigma",))
    dev.timing_reset()
    for _ in range(4): dev.sweep_logexp(None, 0.1, 0.0, 1e-2, M=M, want=())
    ms, n = dev.timing("sweep_gemm")
    print(f"altwalk={alt} persist={ps}: {ms / n:.3f} ms per launch, {M * 4 * (N * N + 2.0 * N) / (ms * 1e-3) / 1e12:.2f} TF", flush=True)
