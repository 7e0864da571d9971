#!/usr/bin/env python3
"""Throughput of LML+gradient evaluations with 1, 2, 3 contexts driven from as many host threads."""
import os
import sys
import threading
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpry_amd import _lib  # noqa: E402

for N, d, reps in ((1024, 8, 60), (4096, 16, 30), (8192, 20, 12)):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
    theta = np.log(np.array([4.0] + [0.3] * d))
    devs = []
    for _ in range(3):
        dv = _lib.Device(0)
        dv.set_option("timing", 0)
        dv.set_train(X, y, np.full(N, 1e-4)); dv.set_theta(3, theta); dv.lml(theta, True)
        devs.append(dv)

    def work(dv, n):
        for i in range(n):
            dv.lml(theta + 1e-3 * i, True)

    for nt in (1, 2, 3, 1, 2):
        ths = [threading.Thread(target=work, args=(devs[k], reps)) for k in range(nt)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        print(f"N={N}: {nt} context(s): {dt / (nt * reps) * 1e3:.3f} ms per evaluation (aggregate), {dt / reps * 1e3:.3f} ms per evaluation of one thread", flush=True)
    for dv in devs: dv.close()
