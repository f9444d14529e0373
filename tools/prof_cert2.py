import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, dcora_amd as da
ds = common.product_dataset("sphere2500")
T = da.chordal_initialization(ds)
X0 = np.zeros((5, 4 * ds.n)); X0[:3] = T
s = da.RbcdSession(ds, num_robots=5, r=5); s.set_X(X0); out = s.run(max_iters=1000, rgrad_tol=0.1); X = s.get_X(); s.close()
Q = da.build_Q_pgo(ds)
for rep in range(3):
    t0 = time.perf_counter(); S = da.dual_certificate(5, 3, ds.n, X, Q); t1 = time.perf_counter()
    psd = da.is_psd(da.Csr.from_scipy(S.to_scipy() + 1e-3 * __import__("scipy.sparse").sparse.identity(S.n, format="csr")), block=4); t2 = time.perf_counter()
    r = da.fast_verification(S, 1e-3, block=4); t3 = time.perf_counter()
    print("dual_certificate %.2f ms   is_psd (incl. python csr) %.2f ms   fast_verification %.2f ms  psd=%s" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), r[0]))
os.environ["DCORA_FACTOR_TIMING"] = "1"
da.fast_verification(S, 1e-3, block=4)
