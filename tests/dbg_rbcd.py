import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common
import dcora_amd as da
from oracle import orc
ds, dso = common.product_dataset("smallGrid3D"), common.oracle_dataset("smallGrid3D")
r = 5
X0 = common.random_point(r, ds.d, ds.n, 1, orc.project_to_manifold)
iters = 14
tr = orc.run_rbcd(dso, X0, num_robots=5, r_min=r, max_iters=iters, staircase=0, rgrad_tol=1e-12)
s = da.RbcdSession(ds, num_robots=5, r=r)
s.set_X(X0)
sel = 0
for it in range(iters):
    c2, gn, bn, nxt = s.iterate(sel)
    lr = s.last_result()
    print(it, sel, tr["selected"][it], "cost rel diff %.3e" % (abs(c2 - tr["cost"][it]) / tr["cost"][it]), "gn %.6f %.6f" % (gn, tr["gradnorm"][it]), lr["outer_iterations"], lr["inner_iterations"], lr["accepted_steps"], lr["tCGStatus"])
    sel = nxt
