"""scratch: the sparse replay of one 100k-lattice agent under the timing switches of sparse_precond.hip (DCORA_SP_EXP)"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common
import dcora_amd as da
from dcora_amd import synth
lat = synth.lattice_se3()
nb, ids, vals = bench.agent_block(lat, 8, 0)
Q = da.build_Q_pgo(lat, n=nb, agent=0, ids=ids, vals=vals)
k = 4 * nb
os.environ["DCORA_PRECOND"] = "sparse"
for r in (5, 7):
    P = da.QuadraticProblem(r, 3, nb, Q, G=np.zeros((r, k)), reg=0.1)
    P.f(np.zeros((r, k)))
    ms, nbytes = P.time_precond(reps=200)
    print(json.dumps({"exp": os.environ.get("DCORA_SP_EXP", "0"), "r": r, "us": ms * 1e3, "MB": nbytes / 1e6, "launches": P.precond_info().get("launches")}), flush=True)
    P.close()
