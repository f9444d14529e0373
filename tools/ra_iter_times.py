"""per-iteration wall times of the 4-robot tiers.pyfg RBCD++ loop (dcora_ra_rbcd_*): percentiles, to see stalls"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
from dcora_amd import datasets
ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
r = 3
X0 = np.zeros((r, ra.k)); X0[:ra.d] = ra.X_odom
s = da.RaRbcdSession(ra, r)
s.set_X(X0)
sel = s.evaluate()[3]
ts = []
for i in range(120):
    t0 = time.perf_counter()
    out = s.iterate(sel)
    ts.append(1e3 * (time.perf_counter() - t0))
    sel = out[3]
ts = np.array(ts)
print("iterations %d: mean %.3f ms, median %.3f, p90 %.3f, max %.3f; over 5 ms: %d; sorted tail %s" % (
    len(ts), ts.mean(), np.median(ts), np.percentile(ts, 90), ts.max(), int((ts > 5).sum()), np.round(np.sort(ts)[-6:], 2)))
