"""scratch: certification of the 100k lattice at its optimum (from the chordal start): where the second goes"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
from dcora_amd import synth
ds = synth.lattice_se3()
T = da.chordal_initialization(ds, device=0)
r, k = 5, 4 * ds.n
X = np.zeros((r, k)); X[:3] = T
Q = da.build_Q_pgo(ds)
P = da.QuadraticProblem(r, 3, ds.n, Q)
opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=50, RTR_tCG_iterations=200, gradnorm_tol=1e-2))
X = opt.optimize(X)
print(opt.getOptResult())
for rep in range(3):
    t0 = time.perf_counter()
    S = da.dual_certificate(r, 3, ds.n, X, Q)
    t1 = time.perf_counter()
    psd, th, v, lm = da.fast_verification(S, 1e-3, block=4)
    t2 = time.perf_counter()
    print("rep %d: dual certificate %.1f ms, fast verification %.1f ms (psd %s)" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), psd), flush=True)
