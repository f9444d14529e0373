"""How far the HIP range-aided path moves the ground truth of the reference's noiseless RA fixtures (a fixed point of
iterate() to 1e-9 in ref tests/testAgent.cpp:157-242, 290-456).  Prints the drift per step beside the oracle's."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da  # noqa: E402
from oracle import orc  # noqa: E402
from test_raslam import ra_path, ra_plain  # noqa: E402

for name in ("range_aided_slam_test_2d", "range_aided_slam_test_3d"):
    ra = da.RADataset(ra_path(name))
    ro = orc.RADataset(ra_plain(name))
    d, n, l, b = ra.d, ra.n, ra.l, ra.b
    print(name, "d n l b", d, n, l, b)
    # centralised problem at the ground truth
    P = da.QuadraticProblem(d, d, n, ra.Q, reg=da.precond_regularization(ra.Q), l=l, b=b)
    Po = orc.Problem(d, d, n, ro.Q, reg=1e-3, l=l, b=b)
    print("  f(gt) hip %.3e oracle %.3e   |rgrad| hip %.3e oracle %.3e" %
          (P.f(ra.gt), Po.f(ro.gt), P.RieGradNorm(ra.gt), np.linalg.norm(Po.rgrad(ro.gt))))
    for tol in (1e-2, 1e-4, 1e-10):
        X = da.QuadraticOptimizer(P, da.ROptParameters(gradnorm_tol=tol)).optimize(ra.gt)
        Xo, _ = Po.optimize(ro.gt, gradnorm_tol=tol)
        print("  optimize tol %.0e: |X - gt|max hip %.3e oracle %.3e" % (tol, np.abs(X - ra.gt).max(), np.abs(Xo - ro.gt).max()))
    for accel in (False, True):
        for prm in (da.ROptParameters(), da.ROptParameters(RTR_iterations=200, RTR_tCG_iterations=200, gradnorm_tol=1e-4)):
            s = da.RaRbcdSession(ra, d, acceleration=accel, params=prm)
            s.set_X(ra.gt)
            dr = []
            for it in range(4):
                for sel in range(s.R):
                    s.iterate(sel)
                    dr.append(np.abs(s.get_X() - ra.gt).max())
            print("  session accel %d tol %.0e: drift per iterate %s" % (accel, prm.gradnorm_tol, " ".join("%.2e" % v for v in dr)))
            s.close() if hasattr(s, "close") else None
