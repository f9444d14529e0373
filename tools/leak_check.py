"""device / host memory across many create-destroy cycles (problems with dense and sparse preconditioners, sessions,
PSD tests), caches cleared each round: free device memory and the process RSS must come back"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import datasets, synth  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def free_mb():
    fr, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    return fr.value / 1e6


def rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS"):
                return int(line.split()[1]) / 1e3
    return 0.0


ds = datasets.product_dataset("sphere2500")
lat = synth.lattice_se3(14, 14, 12)
Q = da.build_Q_pgo(ds)
Ql = da.build_Q_pgo(lat)
import scipy.sparse as sp
S = da.Csr.from_scipy((Q.to_scipy() + 1e-3 * sp.identity(Q.n)).tocsr())
X0 = da.manifold_project(5, 3, ds.n, np.random.default_rng(0).uniform(-1, 1, (5, 4 * ds.n)))
for rnd in range(8):
    for _ in range(5):
        P = da.QuadraticProblem(5, 3, ds.n, Q)      # sparse preconditioner
        P.f(X0)
        P.close()
        P = da.QuadraticProblem(5, 3, lat.n, Ql)    # sparse, smaller
        P.close()
        s = da.RbcdSession(ds, num_robots=5, r=5)   # dense preconditioners
        s.set_X(X0)
        s.run(max_iters=3, rgrad_tol=0.0)
        s.close()
        da.is_psd_device(S, 4)
        da.precond_cache_clear()
        da.chol_cache_clear()
    print("round %d: free device memory %.0f MB, RSS %.0f MB" % (rnd, free_mb(), rss_mb()), flush=True)
