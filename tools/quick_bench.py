"""scratch timing of the RBCD loop on sphere2500 / 5 agents (not a test)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common
import dcora_amd as da

ds = common.product_dataset("sphere2500")
r = 5
rng = np.random.default_rng(1)
t = time.time(); s = da.RbcdSession(ds, num_robots=5, r=r); print("setup s", time.time() - t)
X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, 4 * ds.n)))
s.set_X(X0)
out = s.run(max_iters=20, rgrad_tol=0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t = time.time(); out = s.run(max_iters=K, rgrad_tol=0); dt = time.time() - t
print("iters", out["iters"], "time", dt, "it/s", out["iters"] / dt, "ms/it", 1e3 * dt / out["iters"])
print(out["cost"][:3], out["cost"][-3:], out["gradnorm"][-3:])
print(s.last_result())
