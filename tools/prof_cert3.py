"""where the certification time goes on sphere2500 (dual certificate, fast verification, PSD test)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import datasets  # noqa: E402

ds = datasets.product_dataset("sphere2500")
Q = da.build_Q_pgo(ds)
r = 5
T = da.chordal_initialization(ds)
X0 = np.zeros((r, 4 * ds.n))
X0[:3] = T
s = da.RbcdSession(ds, num_robots=5, r=r)
s.set_X(X0)
s.run(max_iters=300, rgrad_tol=0.1)
X = s.get_X()
s.close()
for rep in range(4):
    if rep == 0:
        da.chol_cache_clear()
    t0 = time.perf_counter()
    S = da.dual_certificate(r, 3, ds.n, X, Q)
    t1 = time.perf_counter()
    psd, th, v, lm = da.fast_verification(S, 1e-3, block=4)
    t2 = time.perf_counter()
    ok, info = da.is_psd_device(S, 4, info=True)
    t3 = time.perf_counter()
    print("rep %d: dual certificate %.2f ms, fast verification %.2f ms (psd %s), is_psd_device alone %.2f ms %s" % (
        rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), psd, 1e3 * (t3 - t2), info), flush=True)
