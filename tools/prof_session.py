import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, dcora_amd as da
name, R = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("sphere2500", 5)
ds = common.product_dataset(name)
s = da.RbcdSession(ds, num_robots=R, r=5); s.close()
for rep in range(3):
    da.precond_cache_clear()
    t0 = time.perf_counter(); s = da.RbcdSession(ds, num_robots=R, r=5); t1 = time.perf_counter(); s.close()
    t2 = time.perf_counter(); s = da.RbcdSession(ds, num_robots=R, r=5); t3 = time.perf_counter(); s.close()
    print("%s / %d agents (%s): session cold %.2f ms, cached %.2f ms" % (name, R, os.environ.get("DCORA_SERIAL_SETUP", "parallel"), 1e3 * (t1 - t0), 1e3 * (t3 - t2)))
