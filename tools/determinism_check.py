"""bitwise repeatability of set-up products and of a solve: the dense inverse of an agent's block built N times with the
cache cleared between builds (preconditioner applied to a fixed vector), and N whole RBCD runs of 30 iterations; any
difference in any bit is reported.  python tools/determinism_check.py [N]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common  # noqa: E402
import dcora_amd as da  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ds = common.product_dataset("sphere2500")
nb, ids, vals = bench.agent_block(ds, 5, 1)
Q = da.build_Q_pgo(ds, n=nb, agent=1, ids=ids, vals=vals)
k, r = 4 * nb, 5
rng = np.random.default_rng(5)
X = common.random_point(r, 3, nb, 3, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
V = rng.standard_normal((r, k))
ref, bad = None, 0
for i in range(N):
    da.precond_cache_clear()
    P = da.QuadraticProblem(r, 3, nb, Q, G=np.zeros((r, k)), reg=da.precond_regularization(Q))
    Z = P.PreCondition(X, V)
    P.close()
    if ref is None:
        ref = Z
    elif not np.array_equal(ref, Z):
        bad += 1
        print("build %d differs: max |dZ| %.3e" % (i, np.abs(ref - Z).max()), flush=True)
print("dense inverse, %d builds: %d differ" % (N, bad), flush=True)
X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
ref, bad = None, 0
for i in range(N):
    da.precond_cache_clear()
    s = da.RbcdSession(ds, 5, r)
    s.set_X(X0)
    out = s.run(max_iters=30, rgrad_tol=0.0)
    Xo = s.get_X()
    s.close()
    sig = (tuple(out["selected"]), Xo.tobytes())
    if ref is None:
        ref = sig
    elif sig != ref:
        bad += 1
        print("run %d differs (selected equal: %s)" % (i, sig[0] == ref[0]), flush=True)
print("RBCD 30 iterations, %d runs: %d differ" % (N, bad), flush=True)
