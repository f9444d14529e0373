"""the first CORA level of tiers.pyfg (r = d = 2, RTR from the odometry start) once: for rocprofv3 --kernel-trace"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import cora_flow, datasets  # noqa: E402

ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
hip = cora_flow.ProductBackend(ra)
P = hip.problem(ra.d)
t = time.perf_counter()
X, f, gn, outer, inner = hip.optimize(P, ra.X_odom)
dt = time.perf_counter() - t
print("f %.6f gradnorm %.3e outer %d tCG %d: %.3f s, %.1f tCG it/s" % (f, gn, outer, inner, dt, inner / dt), flush=True)
P.close()
