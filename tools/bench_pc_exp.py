"""scratch: k_fused_pc in its in-loop form, with and without the partial-sum load + wave sum of <d, H d> in its prologue"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common
import dcora_amd as da
ds = common.product_dataset("sphere2500")
nb, ids, vals = bench.agent_block(ds, 5, 0)
Q = da.build_Q_pgo(ds, n=nb, agent=0, ids=ids, vals=vals)
k = 4 * nb
P = da.QuadraticProblem(5, 3, nb, Q, G=np.zeros((5, k)), reg=0.1)
P.f(np.zeros((5, k)))
for _ in range(3):
    ms, nbytes = P.time_precond(reps=300)
    print("DCORA_PC_EXP=%s: %.3f us" % (os.environ.get("DCORA_PC_EXP", "0"), ms * 1e3))
