"""scratch: Q-apply (Y = X Q + G) on the 100k-pose lattice and on one of its 8 agent blocks"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, dcora_amd as da
from dcora_amd import synth
r = 5
big = synth.lattice_se3()
Qg = da.build_Q_pgo(big)
kg = 4 * big.n
P = da.QuadraticProblem(r, 3, big.n, Qg, G=np.zeros((r, kg)), reg=-1.0)
P.f(np.zeros((r, kg)))
for _ in range(3):
    ms, nbytes = P.time_qapply(reps=100)
    print("lattice100k: %.2f us  %.0f GB/s  %.1f %%" % (ms * 1e3, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
