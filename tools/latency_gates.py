"""Where does a one-point predict with classifier + trust box spend its time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import gpry_oracle as orc
from test_host_mirror_gpu import make_gpr

def per_call(f, n=2000):
    for _ in range(100): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6

bounds, X, y, Xc = orc.synthetic_like_goldens(150, 3, 6000, seed=9)
y = y.copy(); y[X[:, 0] > 1.0] = -np.inf
gpr = make_gpr(bounds, 3, theta=np.log(np.array([4.0, 0.3, 0.3, 0.3])), account_for_inf="SVM", inf_threshold="20s",
               trust_region_factor=1.5, random_state=1)
gpr.append_to_data(X, y, fit_gpr=False)
x = Xc[:1]
dev = gpr.device
gpr.predict(x)
print("mirror predict, device gates      %.1f us" % per_call(lambda: gpr.predict(x, validate=False)))
print("  _sync_gates                     %.1f us" % per_call(lambda: gpr._sync_gates(False)))
print("  _push_affine                    %.1f us" % per_call(lambda: gpr._push_affine()))
print("  Device.predict (gates on dev)   %.1f us" % per_call(lambda: dev.predict(x)), dev.serve_stats())
dev.set_option("predict_gates", 0)
print("  Device.predict (no gates)       %.1f us" % per_call(lambda: dev.predict(x)), dev.serve_stats())
dev.set_option("predict_gates", 1)
dev.set_option("predict_serve", 0)
print("  Device.predict gates, no serve  %.1f us" % per_call(lambda: dev.predict(x)))
dev.set_option("predict_serve", 1)
print("  host _masks                     %.1f us" % per_call(lambda: gpr._masks(x, False, False)))
print("  svc n_sv", gpr.infinities_classifier._svc.support_vectors_.shape)
