"""the coloured simultaneous-update mode on the headline split or (argument `lattice`) on the 100k lattice: block updates/s,
four repeats"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import dcora_amd as da
from dcora_amd import datasets
if "lattice" in sys.argv[1:]:   # the 100k lattice, 8 agents (sparse preconditioner)
    from dcora_amd import synth
    ds = synth.lattice_se3()
    X0 = bench.initial_point(da, ds, 5)
    s = da.RbcdSession(ds, num_robots=8, r=5, acceleration=False)
else:
    ds = datasets.product_dataset("sphere2500")
    X0 = bench.initial_point(da, ds, 5)
    s = da.RbcdSession(ds, num_robots=5, r=5, acceleration=False)
for rep in range(4):
    out = bench.coloured_sweeps(bench.SingleDriver(s), X0, sweeps=40, warm=2)
    print("rep %d: %.0f block updates/s, %.3f ms per sweep" % (rep, out["block_updates_per_s"], out["ms_per_sweep"]), flush=True)
