"""scratch: Q-apply (Y = X Q + G) on the 100k-pose lattice, cache-warm (one set re-read) and HBM-cold (4 sets in turn)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, dcora_amd as da
from dcora_amd import synth
r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
big = synth.lattice_se3()
Qg = da.build_Q_pgo(big)
kg = 4 * big.n
rng = np.random.default_rng(0)
Ps = [da.QuadraticProblem(r, 3, big.n, Qg, G=np.zeros((r, kg)), reg=-1.0) for _ in range(4)]
X = rng.standard_normal((r, kg))
for P in Ps:
    P.f(X)
ms, nbytes = Ps[0].time_qapply(reps=100)
msc = da.time_qapply_rotating(Ps, reps=96)
print("r=%d: warm %.2f us %.1f %%   cold %.2f us %.1f %%" % (r, ms * 1e3, nbytes / ms / 1e6 / 80, msc * 1e3, nbytes / msc / 1e6 / 80))
