"""scratch: the coloured simultaneous-update mode on the headline split (block updates/s), three repeats"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import dcora_amd as da
from dcora_amd import datasets
ds = datasets.product_dataset("sphere2500")
X0 = bench.initial_point(da, ds, 5)
s = da.RbcdSession(ds, num_robots=5, r=5, acceleration=False)
for rep in range(4):
    out = bench.coloured_sweeps(bench.SingleDriver(s), X0, sweeps=40, warm=2)
    print("rep %d: %.0f block updates/s, %.3f ms per sweep" % (rep, out["block_updates_per_s"], out["ms_per_sweep"]), flush=True)
