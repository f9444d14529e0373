"""dense against sparse preconditioner on blocks of a few thousand unknowns: RBCD iterations/s of sphere2500 split into
R agents (k = 10000 / R per block), DCORA_PRECOND forced either way (read when a problem is created)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, dcora_amd as da
from dcora_amd import datasets
ds = datasets.product_dataset("sphere2500")
r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
X0 = bench.initial_point(da, ds, r)
print("r =", r)
for R in (2, 3, 4, 5):
    out = {}
    for kind in ("dense", "sparse"):
        os.environ["DCORA_PRECOND"] = kind
        t0 = time.perf_counter()
        s = da.RbcdSession(ds, num_robots=R, r=r)
        setup = time.perf_counter() - t0
        s.set_X(X0)
        s.run(max_iters=40, rgrad_tol=0.0)
        t0 = time.perf_counter()
        o = s.run(max_iters=200, rgrad_tol=0.0)
        dt = time.perf_counter() - t0
        out[kind] = (200 / dt, setup, float(o["cost"][-1]))
        s.close()
    print("R = %d (k = %5d per block): dense %7.1f it/s (set-up %.3f s)   sparse %7.1f it/s (set-up %.3f s)   costs %.6f / %.6f" % (
        R, 4 * ((ds.n + R - 1) // R), out["dense"][0], out["dense"][1], out["sparse"][0], out["sparse"][1], out["dense"][2], out["sparse"][2]), flush=True)
