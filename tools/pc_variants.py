"""scratch: k_fused_pc back to back (HIP events) on one agent block of sphere2500"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common
import dcora_amd as da
ds = common.product_dataset("sphere2500")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
r = int(sys.argv[2]) if len(sys.argv) > 2 else 5
nb, ids, vals = bench.agent_block(ds, R, 0)
Q = da.build_Q_pgo(ds, n=nb, agent=0, ids=ids, vals=vals)
k = 4 * nb
P = da.QuadraticProblem(r, 3, nb, Q, G=np.zeros((r, k)), reg=0.1)
P.f(np.zeros((r, k)))
ms, nbytes = P.time_precond(reps=300)
print("k=%d r=%d dbg=%s bc=%s: %.2f us  %.0f GB/s" % (k, r, os.environ.get("DCORA_PC_DBG", "0"), os.environ.get("DCORA_SOLVER_BC", "pc"), ms * 1e3, nbytes / ms / 1e6))
