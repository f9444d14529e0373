"""A/B of the dense tCG run as ONE launch (k_tcg_run) against the launches per iteration (k_fused_hess + k_fused_pc):
bitwise equality of the iterates and RBCD iterations/s on the headline split (sphere2500, 5 agents, r = 5)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import common  # noqa: E402
import dcora_amd as da  # noqa: E402

ds = common.product_dataset("sphere2500")
r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
X0 = bench.initial_point(da, ds, r)


def session(form):
    if form:
        os.environ["DCORA_SOLVER_TCG"] = form
    else:
        os.environ.pop("DCORA_SOLVER_TCG", None)
    s = da.RbcdSession(ds, num_robots=5, r=r)
    os.environ.pop("DCORA_SOLVER_TCG", None)
    return s


res = {}
for form in ("launch", None, "launch", None):
    s = session(form)
    s.set_X(X0)
    s.run(max_iters=30, rgrad_tol=0.0)
    s.set_X(X0)
    s.synchronize()
    t0 = time.perf_counter()
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    s.synchronize()
    dt = time.perf_counter() - t0
    X = s.get_X()
    key = form or "run"
    print("%-7s %8.1f RBCD it/s   2f %.12g  |g| %.6g" % (key, iters / dt, out["cost"][-1], out["gradnorm"][-1]), flush=True)
    res.setdefault(key, []).append((out, X))
    s.close()
a, b = res["launch"][0], res["run"][0]
print("selected equal:", np.array_equal(a[0]["selected"], b[0]["selected"]))
print("cost bitwise equal:", np.array_equal(a[0]["cost"], b[0]["cost"]), " max rel diff %.3e" %
      np.max(np.abs(a[0]["cost"] - b[0]["cost"]) / np.abs(a[0]["cost"])))
print("X bitwise equal:", np.array_equal(a[1], b[1]), " max abs diff %.3e" % np.abs(a[1] - b[1]).max())
print("run form repeats itself:", np.array_equal(res["run"][0][1], res["run"][1][1]))
