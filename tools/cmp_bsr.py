"""scratch: Euclidean gradient (Y = X Q + G) of the 100k lattice and of sphere2500 through the block Q-apply, saved for a
bitwise comparison between kernel forms (DCORA_BSR_KERNEL is read once per process)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
from dcora_amd import datasets, synth
out = sys.argv[1]
res = {}
for name, ds in (("lattice", synth.lattice_se3()), ("sphere", datasets.product_dataset("sphere2500")), ("grid2d", datasets.product_dataset("pose_graph_optimization_test_2d"))):
    Q = da.build_Q_pgo(ds)
    k = (ds.d + 1) * ds.n
    for r in (ds.d, 5, 8):
        if r < ds.d:
            continue
        rng = np.random.default_rng(7)
        X = rng.standard_normal((r, k))
        Gm = rng.standard_normal((r, k))
        P = da.QuadraticProblem(r, ds.d, ds.n, Q, G=Gm, reg=-1.0)
        res["%s_r%d_f" % (name, r)] = np.array([P.f(X)])
        res["%s_r%d_g" % (name, r)] = P.EucGrad(X)
        P.close()
np.savez(out, **res)
print("saved", out, len(res))
