"""chordal initialisation of the 100k lattice with the device solves, then the centralised solve from it"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402

ds = synth.lattice_se3()
t = time.perf_counter()
T = da.chordal_initialization(ds, device=0)
print("chordal initialisation on the device: %.1f s" % (time.perf_counter() - t), flush=True)
r, k = 5, 4 * ds.n
X = np.zeros((r, k))
X[:3] = T
Q = da.build_Q_pgo(ds)
t = time.perf_counter()
P = da.QuadraticProblem(r, 3, ds.n, Q)
print("problem %.1f s, 2f(chordal) = %.6e" % (time.perf_counter() - t, 2 * P.f(X)), flush=True)
total = 0.0
for rnd in range(40):
    opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=50, RTR_tCG_iterations=200, gradnorm_tol=1e-2))
    t0 = time.perf_counter()
    X = opt.optimize(X)
    dt = time.perf_counter() - t0
    total += dt
    res = opt.getOptResult()
    print("round %d: %.2f s, 2f %.8e -> %.8e, gradnorm %.3e, outer %d inner %d" % (
        rnd, dt, 2 * res["fInit"], 2 * res["fOpt"], res["gradNormOpt"], res["outer_iterations"], res["inner_iterations"]),
        flush=True)
    if res["gradNormOpt"] < 1e-2:
        break
print("solve from the chordal start: %.2f s" % total, flush=True)
P.close()
s = da.RbcdSession(ds, num_robots=8, r=r)
s.set_X(np.vstack([T, np.zeros((r - 3, k))]))
t0 = time.perf_counter()
out = s.run(max_iters=300, rgrad_tol=0.1)
print("RBCD++ (8 agents) from the chordal start: %d iterations in %.2f s, 2f %.6e -> %.6e, gradnorm %.3e" % (
    out["iters"], time.perf_counter() - t0, out["cost"][0], out["cost"][-1], out["gradnorm"][-1]), flush=True)
s.close()
