import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from test_host_mirror_gpu import make_gpr
from gpry_amd.kernels import clone
g = load_golden("fit")
for v1 in (0, 1, 2):
    gpr = make_gpr(g["f9_bounds"], 0, n_restarts_optimizer=3, random_state=3)
    gpr.device.set_option("trtri_diag_v1", v1)
    gpr.append_to_data(g["f9_X"], g["f9_y"], fit_gpr=True)
    k = clone(gpr.kernel); k.theta = g["f9_theta"]; gpr.kernel_ = k; gpr._invalidate()
    m, s = gpr.predict(g["f9_Xc"], return_std=True)
    print("trtri_diag_v1 =", v1, "max |mean dev| =", np.max(np.abs(m - g["f9_mean"])), "tol", 1e-5 * np.max(np.abs(g["f9_mean"])))
