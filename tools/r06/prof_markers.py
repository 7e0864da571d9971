#!/usr/bin/env python3
"""One refit + acquisition cycle at N = 4096, d = 16 through the mirror classes, for rocprofv3 --marker-trace --kernel-trace with
GPRY_HIP_ROCTX=1 (roctx ranges around the stages of every entry point): prof_markers.py [M]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from gpry_amd.gp_acquisition import NORA
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n_base, d, npts = 4080, 16, 16
bounds, X, y, Xc, truth = bench.synthetic(n_base, d, M)
gpr = bench.make_gpr(bounds); gpr.verbose = 0
gpr.append_to_data(X, y, fit_gpr="simple")
acq = NORA(bounds, sampler="uniform", mc_every=1, verbose=0, devices=[0])
acq.do_MC_sample = lambda gpr, bounds=None, rng=None, sampler=None: (Xc, None, None, None)
rng = np.random.default_rng(2)
X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)
gpr.append_to_data(X_new, truth(X_new), fit_gpr="simple")     # the cycle that is summarised: refit ...
X_new, _, _ = acq.multi_add(gpr, n_points=npts, rng=rng)       # ... and acquisition
th = np.array(gpr.kernel_.theta, dtype=float)
gpr.device.set_option("lml_schedule", 1)
gpr.device.lml_batch(np.tile(th, (6, 1)) + 0.01 * rng.standard_normal((6, d + 1)), True)      # a round of a throughput fit
gpr.device.set_option("lml_schedule", 0)
print("done: N =", gpr.n, "panel form:", acq.stats.get("panel_form"))
