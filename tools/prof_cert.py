"""scratch: breakdown of the certification step on sphere2500 at the certified optimum"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, dcora_amd as da
ds = common.product_dataset("sphere2500")
T = da.chordal_initialization(ds)
r = 5
X0 = np.zeros((r, 4 * ds.n)); X0[:3] = T
s = da.RbcdSession(ds, num_robots=5, r=r)
s.set_X(X0)
t = time.perf_counter(); out = s.run(max_iters=1000, rgrad_tol=0.1); t_run = time.perf_counter() - t
t = time.perf_counter(); s.synchronize(); t_sync = time.perf_counter() - t
t = time.perf_counter(); X = s.get_X(); t_get = time.perf_counter() - t
t = time.perf_counter(); X = s.get_X(); t_get2 = time.perf_counter() - t
print("sync after run %.2f ms, get_X %.2f ms, get_X again %.2f ms" % (1e3 * t_sync, 1e3 * t_get, 1e3 * t_get2))
Q = da.build_Q_pgo(ds)
for rep in range(3):
    t = time.perf_counter(); S = da.dual_certificate(r, ds.d, ds.n, X, Q); t_s = time.perf_counter() - t
    t = time.perf_counter(); psd = da.is_psd(S, block=ds.d + 1); t_psd = time.perf_counter() - t
    t = time.perf_counter(); psd2, th, v, lm = da.fast_verification(S, 1e-3, block=ds.d + 1); t_fv = time.perf_counter() - t
    print("run %.1f ms (%d it) get_X %.2f ms | dual_certificate %.2f ms, is_psd %.2f ms, fast_verification %.2f ms" %
          (1e3 * t_run, out["iters"], 1e3 * t_get, 1e3 * t_s, 1e3 * t_psd, 1e3 * t_fv), psd, psd2, flush=True)
