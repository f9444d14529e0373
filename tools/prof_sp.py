"""scratch: one problem, a few sparse-preconditioner applications (for rocprofv3 --kernel-trace)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common, dcora_amd as da
from dcora_amd import synth
os.environ["DCORA_PRECOND"] = "sparse"
lat = synth.lattice_se3()
nb, ids, vals = bench.agent_block(lat, 8, 0)
Q = da.build_Q_pgo(lat, n=nb, agent=0, ids=ids, vals=vals)
k = 4 * nb
P = da.QuadraticProblem(5, 3, nb, Q, G=np.zeros((5, k)), reg=0.1)
P.f(np.zeros((5, k)))
print(P.time_precond(reps=int(sys.argv[1]) if len(sys.argv) > 1 else 2))
