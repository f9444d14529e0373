"""long multi-rank runs of the HEADLINE split on one GPU (sphere2500 / 5 agents over 2 and 4 ranks, both transports, device
and host waits): the one-launch tCG runs give up routinely there (other ranks' waiting kernels hold compute units) and
the solves continue on the launches -- blocks and iterates must stay bitwise those of the single session"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
import dcora_amd as da
import test_exchange_gpu as T
ds = common.product_dataset("sphere2500")
r = 5
X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
for world, transport, wait, iters in ((4, "staged", "device", 300), (2, None, "device", 300), (4, None, "host", 200)):
    cost, gn, sel, X = T.single(da, ds, 5, r, iters, "greedy", X0)
    with tempfile.TemporaryDirectory() as tmp:
        res = T.run_ranks(tmp, world, "sphere2500", 5, r, iters, "greedy", X0, transport, wait)
    ok = all(np.array_equal(o["selected"], sel) and np.array_equal(o["X"], X) for o in res)
    print("ranks %d transport %s wait %s, %d iterations: blocks and iterates bitwise equal to the single session: %s" % (world, transport, wait, iters, ok), flush=True)
