import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
from dcora_amd import synth
def P(*a):
    print("%.1f" % time.time(), *a, flush=True)
ds = synth.lattice_se3(); P("lattice")
R, r = 8, 5
rng = np.random.default_rng(20250310)
X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n))); P("X0")
for rep in range(2):
    s = da.RbcdSession(ds, num_robots=R, r=r); P("session", rep, da.precond_cache_info())
    s.set_X(X0)
    out = s.run(max_iters=4 if rep == 0 else 40, rgrad_tol=0.1); P("run", out["iters"], out["cost"][-1])
    X = s.get_X(); s.close(); P("closed")
Q = da.build_Q_pgo(ds); P("Q")
S = da.dual_certificate(r, ds.d, ds.n, X, Q); P("S")
psd, theta, v, lmin = da.fast_verification(S, 1e-3, block=ds.d + 1); P("fv", psd, theta, lmin)
