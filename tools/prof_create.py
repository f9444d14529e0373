import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common, dcora_amd as da
ds = common.product_dataset("sphere2500")
nb, ids, vals = bench.agent_block(ds, 5, 0)
Q = da.build_Q_pgo(ds, n=nb, agent=0, ids=ids, vals=vals)
P = da.QuadraticProblem(5, 3, nb, Q)
os.environ["DCORA_INIT_TIMING"] = "1"
for r in (5, 6, 5):
    t0 = time.perf_counter(); P2 = da.QuadraticProblem(r, 3, nb, Q); t1 = time.perf_counter()
    print("create r=%d: %.3f ms" % (r, 1e3 * (t1 - t0)), flush=True)
    t0 = time.perf_counter(); P2.close(); print("close %.3f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
