#!/usr/bin/env python3
"""Section stamps of the panel step (diagnostic build, tools/r05/build_stamps.sh): python3 tools/r05/panel_stamps.py N
Prints, per stamped workgroup (diagonal, first off-diagonal, ninth) and wave, the mean s_memtime ticks (= core cycles on
gfx950) between the stamps of a step, over the steps of the last factorisation."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["GPRY_HIP_LIB"] = os.path.join(ROOT, "tools", "r05", "libgpry_hip_stamps.so")
from gpry_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = 8
dev = _lib.Device(0)
dev.set_option("factor_pipeline", 0)
rng = np.random.default_rng(N)
X = rng.uniform(0, 1, (N, d)); y = rng.standard_normal(N)
dev.set_train(X, y, np.full(N, 1e-4))
dev.set_theta(3, np.log(np.array([4.0] + [0.3] * d)))
for _ in range(3):
    assert dev.factorize() == 0
lib = _lib.load_library()
STEPS, SLOTS = 160, 8
buf = np.zeros(STEPS * 3 * 8 * SLOTS, dtype=np.int64)
lib.gpry_debug_panel_stamps.restype = C.c_int
lib.gpry_debug_panel_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.gpry_debug_panel_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
s = buf.reshape(STEPS, 3, 8, SLOTS)
abuf = np.zeros(STEPS * 3 * 8 * 4, dtype=np.int64)
lib.gpry_debug_panel_acc.restype = C.c_int
lib.gpry_debug_panel_acc.argtypes = [C.c_void_p, C.c_int]
assert lib.gpry_debug_panel_acc(abuf.ctypes.data_as(C.c_void_p), abuf.size) == 0
acc = abuf.reshape(STEPS, 3, 8, 4)
nsteps = (N + 127) // 128 * 128 // 64
names = ["start->D,Pt in LDS", "update of D / commit", "factor chain up to own chol16", "own-row tasks", "wait at barrier", "store"]
for wg, wname in enumerate(["diagonal workgroup", "first off-diagonal", "workgroup 8"]):
    print(wname)
    for w in range(8):
        rows = []
        for st in range(1, nsteps - (0 if wg == 0 else 1 if wg == 1 else 8)):
            v = s[st, wg, w]
            if v[0] == 0 or v[6] == 0:
                continue
            rows.append(np.diff(v[:7]))
        if not rows:
            continue
        m = np.mean(np.array(rows), axis=0)
        tot = m.sum()
        print(f"  wave {w} ({['row 0', 'row 1', 'row 2', 'row 3', 'worker', 'worker', 'worker', 'worker'][w]}): " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, m)) + f"; total {tot:.0f} cycles ({len(rows)} steps)")
    if wg > 0:
        sel = [st for st in range(1, nsteps - 8) if s[st, wg, 0, 0]]
        a = acc[sel, wg].mean(axis=0)
        print("  own-row tasks per wave [cycles in U, M, T, tasks pulled]: " + "; ".join(f"w{w}: {a[w, 0]:.0f} {a[w, 1]:.0f} {a[w, 2]:.0f} {a[w, 3]:.1f}" for w in range(8)))
    # chain view: start of step to the end of wave 3's chol16 (the factor's end), and to the store's end
    ends = [(s[st, wg, 3, 3] - s[st, wg, 0, 0], s[st, wg, 0, 6] - s[st, wg, 0, 0]) for st in range(1, nsteps - 8) if s[st, wg, 0, 0] and s[st, wg, 3, 3]]
    if ends:
        e = np.mean(np.array(ends), axis=0)
        print(f"  start -> factor done {e[0]:.0f} cycles; start -> stored {e[1]:.0f} cycles")
