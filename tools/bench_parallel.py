"""scratch measurement: block updates per second of coloured simultaneous ticks against one-after-the-other updates
(same non-accelerated agents, same order), on one GPU"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common  # noqa: E402
import dcora_amd as da  # noqa: E402


def main():
    name, R, r = (sys.argv[1], int(sys.argv[2]), 5) if len(sys.argv) > 2 else ("sphere2500", 5, 5)
    sweeps = 40
    ds = common.product_dataset(name)
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, (ds.d + 1) * ds.n)))
    par = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    seq = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    col, nc = par.colours()
    sets = [np.flatnonzero(col == c).astype(np.int32) for c in range(nc)]
    print("colours", col.tolist())
    for s, mode in ((par, "par"), (seq, "seq"), (par, "par"), (seq, "seq")):
        s.set_X(X0)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(sweeps):
            for S in sets:
                if mode == "par":
                    s.iterate_set(S)
                else:
                    for a in S:
                        s.phase_selected(int(a))
        s.synchronize()
        dt = time.perf_counter() - t0
        c2 = s.evaluate()[0]
        print("%s: %d block updates in %.1f ms -> %.0f updates/s, %.3f ms per sweep, 2f = %.9f" %
              (mode, sweeps * R, dt * 1e3, sweeps * R / dt, dt * 1e3 / sweeps, c2), flush=True)


main()
