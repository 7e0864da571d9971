#!/usr/bin/env python3
"""Host-side masks (libsvm + numpy) vs device gates for one NORA-sized candidate pool."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle import gpry_oracle as orc  # noqa: E402
from test_host_mirror_gpu import make_gpr  # noqa: E402

N, d, M = 2000, 8, 1000000
bounds, X, y, Xc = orc.synthetic_like_goldens(N, d, M, seed=31)
y = y.copy()
y[X[:, 0] > 1.0] = -np.inf
gpr = make_gpr(bounds, 3, theta=np.log(np.array([4.0] + [0.3] * d)), account_for_inf="SVM",
               inf_threshold="20s", trust_region_factor=1.5, random_state=1)
gpr.append_to_data(X, y, fit_gpr=False)
print("training points", gpr.n, "of", N, "; support vectors", len(gpr.infinities_classifier.device_params()[0]))
t0 = time.perf_counter()
host = gpr._masks(Xc, False, False)
print(f"host masks for {M} candidates: {time.perf_counter() - t0:.2f} s")
gpr._ensure_factor(); gpr._push_affine()
gpr._push_gates()
gpr.device.sweep_logexp(Xc, 0.3, gpr.y_max, gpr.noise_level, want=())
gpr.device.timing_reset()
t0 = time.perf_counter()
out = gpr.device.sweep_logexp(None, 0.3, gpr.y_max, gpr.noise_level, M=M, want=("y",))
print(f"sweep with device gates: {time.perf_counter() - t0:.3f} s; gates kernel {gpr.device.timing('gates')[0]:.2f} ms")
print("verdicts differing from the host:", int((np.isneginf(out['y']) != (host != 0)).sum()))
