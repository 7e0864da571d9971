#!/usr/bin/env python3
"""One shape of the batched objective in the throughput schedule for rocprofv3 --kernel-trace: prof_tp.py N d B [streams] [block] [tail]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 2
blk = int(sys.argv[5]) if len(sys.argv) > 5 else 512
tail = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1); y = (y - y.mean()) / y.std()
dv = _lib.Device(0)
dv.set_train(X, y, np.full(N, 1e-4))
base = np.log(np.array([2.0] + [0.5] * d)); dv.set_theta(3, base)
th = base + rng.uniform(-0.3, 0.3, (B, d + 1))
dv.set_option("lml_schedule", 1); dv.set_option("lml_streams", streams); dv.set_option("tp_block", blk); dv.set_option("tp_tail", tail)
dv.lml_batch(th, True)
t0 = time.perf_counter()
for _ in range(3):
    dv.lml_batch(th, True)
print(f"N={N} d={d} B={B} streams={streams}: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per call")
dv.close()
