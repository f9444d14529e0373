"""scratch: where fast_verification's time goes on sphere2500 at the optimum"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
from dcora_amd import datasets
ds = datasets.product_dataset("sphere2500")
Q = da.build_Q_pgo(ds)
r = 5
T = da.chordal_initialization(ds)
X0 = np.zeros((r, 4 * ds.n)); X0[:3] = T
s = da.RbcdSession(ds, num_robots=5, r=r); s.set_X(X0); s.run(max_iters=300, rgrad_tol=0.1); X = s.get_X(); s.close()
S = da.dual_certificate(r, 3, ds.n, X, Q)
M = da.Csr.from_scipy((S.to_scipy() + 1e-3 * sp.identity(S.n)).tocsr())
for rep in range(3):
    t0 = time.perf_counter(); a = da.fast_verification(S, 1e-3, block=4); t1 = time.perf_counter()
    b = da.is_psd_device(M, 4); t2 = time.perf_counter()
    c = da.fast_verification(S, 1e-3, block=4, device=0); t3 = time.perf_counter()
    print("fast_verification %.2f ms (psd %s), is_psd_device(S + eta I) %.2f ms (%s), again %.2f ms" % (1e3 * (t1 - t0), a[0], 1e3 * (t2 - t1), b, 1e3 * (t3 - t2)))
