"""scratch: is the RA session deterministic run to run? (tiers, 4 robots, r = 3)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
from test_raslam import ra_path
name = sys.argv[1] if len(sys.argv) > 1 else "tiers"
ra = da.RADataset(ra_path(name))
r = 3 if ra.d == 2 else 4
if name == "tiers":
    X0 = np.zeros((r, ra.k)); X0[:ra.d] = ra.X_odom
else:
    rng = np.random.default_rng(5)
    lift = np.linalg.qr(rng.standard_normal((r, ra.d)))[0]
    X0 = da.manifold_project(r, ra.d, ra.n, lift @ ra.gt + 0.05 * rng.standard_normal((r, ra.k)), l=ra.l, b=ra.b)
Xs = []
for rep in range(6):
    s = da.RaRbcdSession(ra, r, acceleration=True, restart_interval=4)
    s.set_X(X0)
    out = s.run(max_iters=12, rgrad_tol=0.0)
    Xs.append((s.get_X(), out["gradnorm"].copy()))
    s.close()
for rep in range(1, 6):
    print(rep, "max |dX| vs run 0:", np.max(np.abs(Xs[rep][0] - Xs[0][0])), "gradnorm equal:", np.array_equal(Xs[rep][1], Xs[0][1]), Xs[rep][1][-3:])
