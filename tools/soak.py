"""long runs of the session loops (no assertion beyond finiteness / monotone tail): 20 000 accelerated RBCD++ iterations on
sphere2500 / 5 agents and 2 000 coloured sweeps; used to look for drift or pacing stalls"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import datasets  # noqa: E402

ds = datasets.product_dataset("sphere2500")
T = da.chordal_initialization(ds)
X0 = np.zeros((5, 4 * ds.n))
X0[:3] = T
s = da.RbcdSession(ds, num_robots=5, r=5)
s.set_X(X0)
t = time.perf_counter()
out = s.run(max_iters=20000, rgrad_tol=0.0)
dt = time.perf_counter() - t
c = out["cost"]
print("20000 iterations in %.1f s (%.0f it/s): 2f %.10f -> %.10f, gradnorm %.3e, finite %s, max increase over the last 10000: %.3e"
      % (dt, 20000 / dt, c[0], c[-1], out["gradnorm"][-1], bool(np.all(np.isfinite(c))), float(np.max(np.diff(c[10000:])))),
      flush=True)
s.close()
s = da.RbcdSession(ds, num_robots=5, r=5, acceleration=False)
s.set_X(X0)
col, nc = s.colours()
t = time.perf_counter()
for sweep in range(2000):
    for cc in range(nc):
        s.iterate_set(np.flatnonzero(col == cc).astype(np.int32))
c2, g, bn, nxt = s.evaluate()
print("2000 coloured sweeps in %.1f s: 2f %.10f gradnorm %.3e" % (time.perf_counter() - t, c2, g), flush=True)
s.close()
