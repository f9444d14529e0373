"""scratch: where the time of the CORA flow goes on the GPU"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, cora_flow, dcora_amd as da
name = sys.argv[1] if len(sys.argv) > 1 else "single_drone"
ra = da.RADataset(os.path.join(common.DATA, name + ".pyfg.gz"))
print("k", ra.k, "nnz", ra.Q.nnz, flush=True)
t = time.time(); hip = cora_flow.ProductBackend(ra); print("reg %.3f s" % (time.time() - t), hip.reg, flush=True)
class Timed(cora_flow.ProductBackend):
    pass
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.time(); out = f(*a, **k); print("  %-12s %.3f s" % (name, time.time() - t), flush=True); return out
    setattr(obj, name, g)
for nm in ["problem", "optimize", "certificate", "escape", "project"]:
    wrap(hip, nm)
t = time.time()
out = cora_flow.cora(hip, ra.X_odom, ra.d)
print("total %.3f s" % (time.time() - t))
for lv in out["levels"]:
    print(lv)
P = cora_flow.ProductBackend.problem(hip, ra.d)
print(P.precond_info())
