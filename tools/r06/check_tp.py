#!/usr/bin/env python3
"""Throughput schedule of gpry_lml_batch (option "lml_schedule" = 1) against the latency schedule and against itself:
  * every theta: throughput vs latency within rounding (LML rel, gradient rel to its largest entry);
  * B-invariance: theta i evaluated alone, in a batch of 3 and in the full batch gives the same bits;
  * the factor L of gpry_factorize with the column blocks of the throughput schedule ("chol_tp_segments") = the latency one, bit for bit;
  * timing of both schedules.
usage: check_tp.py N d B [streams] [tp_block] [tp_tail]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpry_amd import _lib
N, d, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 2
blk = int(sys.argv[5]) if len(sys.argv) > 5 else 512
tail = int(sys.argv[6]) if len(sys.argv) > 6 else 1024
left = int(sys.argv[7]) if len(sys.argv) > 7 else 0
kid = 3
rng = np.random.default_rng(0)
X = rng.uniform(size=(N, d)); y = np.sin(3 * X).sum(1); y = (y - y.mean()) / y.std()
dv = _lib.Device(0)
dv.set_train(X, y, np.full(N, 1e-4))
base = np.log(np.array([2.0] + [0.5] * d)); dv.set_theta(kid, base)
th = base + rng.uniform(-0.3, 0.3, (B, d + 1))
dv.set_option("lml_streams", streams); dv.set_option("tp_block", blk); dv.set_option("tp_tail", tail); dv.set_option("tp_left", left)
if N > 4096:
    dv.set_option("lml_batch", 8192)        # (the default stops at 4096, where the host's thread farm takes over)

def best(f, reps=3):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)

dv.set_option("lml_schedule", 0)
l0, g0, i0 = dv.lml_batch(th, True)
t_lat = best(lambda: dv.lml_batch(th, True))
dv.set_option("lml_schedule", 1)
l1, g1, i1 = dv.lml_batch(th, True)
t_tp = best(lambda: dv.lml_batch(th, True))
rel_l = np.max(np.abs(l1 - l0) / np.abs(l0))
rel_g = np.max(np.abs(g1 - g0).max(1) / np.abs(g0).max(1))
ok_inv = True
for i in (0, B // 2, B - 1):
    la, ga, _ = dv.lml_batch(th[i:i + 1], True)
    ok_inv &= (la[0] == l1[i]) and np.array_equal(ga[0], g1[i])
    if B >= 3:
        j = min(i, B - 3)
        lb, gb, _ = dv.lml_batch(th[j:j + 3], True)
        ok_inv &= (lb[i - j] == l1[i]) and np.array_equal(gb[i - j], g1[i])
# the same call again, from scratch sets full of NaNs: deterministic, and independent of what the sets held?
dv.set_option("panel_debug", 128)
l2, g2, _ = dv.lml_batch(th, True)
dv.set_option("panel_debug", 0)
ok_det = np.array_equal(l1, l2) and np.array_equal(g1, g2)
# factor bits
dv.set_option("lml_schedule", 0)
ok_L = None
if N <= 4096:
    dv.set_option("chol_tp_segments", 0); dv.set_theta(kid, th[0]); dv.factorize(); L0, V0, a0 = dv.get_factor()
    dv.set_option("chol_tp_segments", 1); dv.set_theta(kid, th[0]); dv.factorize(); L1, V1, a1 = dv.get_factor()
    dv.set_option("chol_tp_segments", 0)
    ok_L = bool(np.array_equal(L0, L1))
print(f"N={N} d={d} B={B} streams={streams} block={blk} tail={tail} left={left}: latency {t_lat * 1e3:.3f} ms | throughput {t_tp * 1e3:.3f} ms "
      f"(x{t_lat / t_tp:.2f}); lml rel {rel_l:.2e} grad rel {rel_g:.2e}; B-invariant {ok_inv}; deterministic {ok_det}; "
      f"L(tp segments) == L(latency) {ok_L}; info {int(np.abs(i0).max())}/{int(np.abs(i1).max())}", flush=True)
dv.close()
