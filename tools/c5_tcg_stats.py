"""scratch: how the tCG runs of the C5 loop end (outer / inner iterations per RBCD iteration, last status)"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
from dcora_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "lattice"
if name == "lattice":
    ds, R = synth.lattice_se3(), 8
else:
    import common
    ds, R = common.product_dataset(name), int(sys.argv[2])
r = 5
rng = np.random.default_rng(20250310)
X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, 4 * ds.n)))
s = da.RbcdSession(ds, num_robots=R, r=r)
s.set_X(X0)
sel = 0
outer, inner, status, acc = [], [], collections.Counter(), []
for it in range(60):
    c2, gn, bn, sel = s.iterate(sel)
    res = s.last_result()
    outer.append(res["outer_iterations"]); inner.append(res["inner_iterations"]); status[res["tCGStatus"]] += 1; acc.append(res["accepted_steps"])
print("outer per RBCD iteration: mean %.2f, inner (tCG) per RBCD iteration: mean %.2f, per outer %.2f" % (np.mean(outer), np.mean(inner), np.sum(inner) / np.sum(outer)))
print("status of the last tCG run of each iteration (0 negative curvature, 1 boundary, 2 linear, 3 superlinear, 4 max inner):", dict(status))
print("accepted steps per RBCD iteration: mean %.2f" % np.mean(acc), acc[:30])
print(res)
