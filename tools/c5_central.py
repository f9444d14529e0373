"""the whole 100k lattice as ONE problem (k = 400 000): creation (central partitioned-inverse preconditioner built from
the device factorisation), preconditioner application, a few RTR iterations"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402

t0 = time.perf_counter()
ds = synth.lattice_se3()
Q = da.build_Q_pgo(ds)
print("lattice + Q %.1f s" % (time.perf_counter() - t0), flush=True)
r = 5
k = 4 * ds.n
t0 = time.perf_counter()
P = da.QuadraticProblem(r, 3, ds.n, Q)
print("problem created in %.1f s: %s" % (time.perf_counter() - t0, P.precond_info()), flush=True)
ms, nbytes = P.time_precond(reps=20)
print("preconditioner application %.1f us, %.1f MB" % (1e3 * ms, nbytes / 1e6), flush=True)
rng = np.random.default_rng(20250310)
X0 = da.manifold_project(r, 3, ds.n, rng.uniform(-1, 1, (r, k)))
total = 0.0
for rnd in range(40):
    opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=50, RTR_tCG_iterations=200, gradnorm_tol=1e-2))
    t0 = time.perf_counter()
    X = opt.optimize(X0)
    dt = time.perf_counter() - t0
    total += dt
    res = opt.getOptResult()
    print("round %d: %.2f s, 2f %.8e -> %.8e, gradnorm %.3e -> %.3e, outer %d inner %d" % (
        rnd, dt, 2 * res["fInit"], 2 * res["fOpt"], res["gradNormInit"], res["gradNormOpt"],
        res["outer_iterations"], res["inner_iterations"]), flush=True)
    X0 = X
    if res["gradNormOpt"] < 1e-2:
        break
print("solve total %.2f s" % total, flush=True)
t0 = time.perf_counter()
S = da.dual_certificate(r, 3, ds.n, X, Q)
t1 = time.perf_counter()
psd, theta, v, lmin = da.fast_verification(S, 1e-3, block=4)
t2 = time.perf_counter()
print("dual certificate %.2f s, fast verification %.2f s: psd %s theta %.4e lambda_min %.4e" % (
    t1 - t0, t2 - t1, psd, theta, lmin), flush=True)
gap = da.suboptimality_gap(r, 3, ds.n, X, psd, 1e-3, lmin)
print("gap", gap, flush=True)
np.save(os.path.join(ROOT, "gpurun_out", "c5_central_X.npy"), X[:, :4000])
P.close()
