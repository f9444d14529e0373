"""scratch: per-iteration relative difference of the tiers RBCD trace, GPU session vs oracle loop"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
import test_ra_session as T
from test_raslam import ra_path
import common
from oracle import orc
ra = da.RADataset(ra_path("tiers"))
r = 3
X0 = np.zeros((r, ra.k)); X0[:ra.d] = ra.X_odom
for inner in (50, 10, 3):
    opt = dict(RTR_iterations=3, RTR_tCG_iterations=inner, gradnorm_tol=1e-2)
    prm = da.ROptParameters(**opt)
    s = da.RaRbcdSession(ra, r, acceleration=True, restart_interval=4, params=prm)
    s.set_X(X0)
    out = s.run(max_iters=6, rgrad_tol=0.0)
    Xo, tr = T._oracle_loop(da, orc, ra, X0, r, 6, True, 4, opt)
    print("tCG", inner, "rel diff of cost per iteration:", np.abs(out["cost"] - tr[:, 1]) / np.abs(tr[:, 1]))
