"""workload for the rocprofv3 --pmc passes: a few RBCD iterations + the standalone roofline kernels"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common
import dcora_amd as da
import bench
ds = common.product_dataset("sphere2500")
X0 = bench.initial_point(da, ds, 5)
s = da.RbcdSession(ds, num_robots=5, r=5)
s.set_X(X0)
s.run(max_iters=30, rgrad_tol=0.0)
main, extra = bench.roofline(da, ds, 5, 5)
print(main["avg_launch_us"], extra["qapply_lattice100k"].get("avg_launch_us"))
