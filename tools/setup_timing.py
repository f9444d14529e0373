import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, common, dcora_amd as da
ds = common.product_dataset("sphere2500")
s = da.RbcdSession(ds, num_robots=5, r=5); s.close()
for rep in range(3):
    da.precond_cache_clear()
    os.environ["DCORA_INIT_TIMING"] = "1"
    t0 = time.perf_counter()
    s = da.RbcdSession(ds, num_robots=5, r=5)
    print("session create %.2f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
    s.close()
