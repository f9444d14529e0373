"""differential fuzzing of the solver paths: the same random problem solved by the default kernels and by the generic
path (DCORA_SOLVER=generic), with the dense and with the sparse preconditioner -- iteration counts, cost and iterate must
agree; cost / gradient norm are also checked against scipy"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for case in range(ncases):
    lo, hi = (int(os.environ.get("FUZZ_LO", 2)), int(os.environ.get("FUZZ_HI", 14)))
    dims = tuple(int(x) for x in rng.integers(lo, hi, 3))
    r = int(rng.integers(3, 9))
    ds = synth.lattice_se3(*dims, seed=int(rng.integers(1, 1 << 30)))
    n, k = ds.n, 4 * ds.n
    Q = da.build_Q_pgo(ds)
    A = Q.to_scipy()
    X0 = da.manifold_project(r, 3, n, rng.uniform(-1, 1, (r, k)))
    G = rng.standard_normal((r, k)) * float(rng.choice([0.0, 1.0, 30.0]))
    withG = bool(np.any(G))
    f = lambda Y: 0.5 * float(np.sum((A @ Y.T).T * Y)) + float(np.sum(Y * G))
    outs = {}
    for tag, env in (("default", {}), ("generic", {"DCORA_SOLVER": "generic"}), ("sparse", {"DCORA_PRECOND": "sparse"}),
                     ("generic+sparse", {"DCORA_SOLVER": "generic", "DCORA_PRECOND": "sparse"})):
        for kk in ("DCORA_SOLVER", "DCORA_PRECOND"):
            os.environ.pop(kk, None)
        os.environ.update(env)
        P = da.QuadraticProblem(r, 3, n, Q, G=G if withG else None)
        opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=3, RTR_tCG_iterations=30, gradnorm_tol=1e-2))
        X = opt.optimize(X0)
        res = opt.getOptResult()
        outs[tag] = (X, res)
        P.close()
        e0 = abs(res["fInit"] - f(X0)) / max(1.0, abs(f(X0)))
        e1 = abs(res["fOpt"] - f(X)) / max(1.0, abs(f(X0)))
        if e0 > 1e-10 or e1 > 1e-10:
            bad += 1
            print("BOOKKEEPING", dims, r, tag, e0, e1, flush=True)
    for kk in ("DCORA_SOLVER", "DCORA_PRECOND"):
        os.environ.pop(kk, None)
    Xr, rr = outs["default"]
    line = "case %2d dims %s n %4d r %d G %s:" % (case, dims, n, r, withG)
    for tag in ("generic", "sparse", "generic+sparse"):
        X, res = outs[tag]
        dx = np.linalg.norm(X - Xr) / np.linalg.norm(Xr)
        same = (res["outer_iterations"], res["inner_iterations"]) == (rr["outer_iterations"], rr["inner_iterations"])
        line += " %s dX %.1e df %.1e its %s;" % (tag, dx, abs(res["fOpt"] - rr["fOpt"]) / max(1, abs(rr["fOpt"])), same)
        if dx > 1e-6 or not same:
            bad += 1
            line += " <== MISMATCH"
    print(line, flush=True)
print("mismatches:", bad)
