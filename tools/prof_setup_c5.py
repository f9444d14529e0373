"""set-up of one agent of the 100k lattice and of the 8-agent session (DCORA_INIT_TIMING / DCORA_FACTOR_TIMING laps on
stderr)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402

lat = synth.lattice_se3()
nb, ids, vals = bench.agent_block(lat, 8, 0)
Q = da.build_Q_pgo(lat, n=nb, agent=0, ids=ids, vals=vals)
t = time.perf_counter()
P = da.QuadraticProblem(5, 3, nb, Q, G=np.zeros((5, 4 * nb)), reg=0.1)
print("one agent: %.3f s" % (time.perf_counter() - t), flush=True)
P.close()
da.precond_cache_clear()
t = time.perf_counter()
s = da.RbcdSession(lat, num_robots=8, r=5)
print("session of 8: %.3f s" % (time.perf_counter() - t), flush=True)
s.close()
