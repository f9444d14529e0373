"""where the certification milliseconds of the headline's `ms_to_certified_optimum` go: the steps after the RBCD loop of
dcora_amd/driver.py timed one by one (sphere2500, 5 agents, chordal start, cold caches).  python tools/cert_timing.py"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import dcora_amd as da  # noqa: E402

ds = common.product_dataset("sphere2500")
r = 5
T = da.chordal_initialization(ds)
X0 = np.zeros((r, 4 * ds.n)); X0[:3] = T
Q = da.build_Q_pgo(ds)
for rep in range(3):
    da.precond_cache_clear()
    if rep < 2:
        da.chol_cache_clear()
    import threading
    t = [time.perf_counter()]
    prep = threading.Thread(target=da.cert_prepare, args=(Q, 3, ds.n), kwargs=dict(block=4), daemon=True)
    prep.start()
    s = da.RbcdSession(ds, num_robots=5, r=r)
    s.set_X(X0); t.append(time.perf_counter())
    out = s.run(max_iters=1000, rgrad_tol=0.1); t.append(time.perf_counter())
    Xopt = s.get_X(); t.append(time.perf_counter())
    S = da.dual_certificate(r, 3, ds.n, Xopt, Q); t.append(time.perf_counter())
    prep.join(); t.append(time.perf_counter())
    psd, theta, v, lmin = da.fast_verification(S, 1e-3, block=4); t.append(time.perf_counter())
    s.close(); t.append(time.perf_counter())
    names = ["setup", "rbcd (%d its)" % out["iters"], "get_X", "dual_certificate", "join", "fast_verification", "close"]
    print("rep %d (chol cache %s): " % (rep, "cold" if rep < 2 else "warm") +
          ", ".join("%s %.2f ms" % (n, 1e3 * (b - a)) for n, a, b in zip(names, t[:-1], t[1:])) + "; psd %s" % psd, flush=True)
