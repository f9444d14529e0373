"""scratch: the dense preconditioner kernel in three contexts (for rocprofv3 --kernel-trace --stats):
   mode micro: back-to-back applications (what bench.py's roofline times); mode solve: inside one agent's long RTR solve;
   mode rbcd: inside the RBCD loop (agents alternate)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common, dcora_amd as da
mode = sys.argv[1]
ds = common.product_dataset("sphere2500")
r = 5
if mode in ("micro", "solve"):
    nb, ids, vals = bench.agent_block(ds, 5, 0)
    Q = da.build_Q_pgo(ds, n=nb, agent=0, ids=ids, vals=vals)
    k = 4 * nb
    P = da.QuadraticProblem(r, 3, nb, Q, G=np.zeros((r, k)), reg=0.1)
    if mode == "micro":
        P.f(np.zeros((r, k)))
        print(P.time_precond(reps=2000))
    else:
        rng = np.random.default_rng(0)
        X0 = da.manifold_project(r, 3, nb, rng.uniform(-1, 1, (r, k)))
        opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=60, RTR_tCG_iterations=200, gradnorm_tol=1e-12))
        opt.optimize(X0)
        print(opt.getOptResult())
else:
    X0 = bench.initial_point(da, ds, r)
    s = da.RbcdSession(ds, num_robots=5, r=r)
    s.set_X(X0)
    out = s.run(max_iters=600, rgrad_tol=0.0)
    print(out["cost"][-1])
