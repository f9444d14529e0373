"""long multi-rank run on one GPU: 4 ranks, torus3D / 8 agents, 1500 greedy iterations through the library's exchange,
compared with the single session (same blocks, iterates bit for bit) -- the parity slots and sequence numbers of the
exchange over many rounds and restarts"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
import dcora_amd as da  # noqa: E402
import test_exchange_gpu as tx  # noqa: E402

name, R, world, iters = "torus3D", 8, 4, 1500
ds = common.product_dataset(name)
rng = np.random.default_rng(7)
X0 = da.manifold_project(5, ds.d, ds.n, rng.uniform(-1, 1, (5, 4 * ds.n)))
for transport in (None, "staged"):
    with tempfile.TemporaryDirectory() as tmp:
        res = tx.run_ranks(tmp, world, name, R, 5, iters, "greedy", X0, transport)
        cost, gn, sel, X = tx.single(da, ds, R, 5, iters, "greedy", X0)
        r0 = res[0]
        same_sel = bool(np.array_equal(r0["selected"], sel))
        print("transport %s (%s): same blocks %s, max |cost - single| / cost %.2e, iterates bitwise %s, posts %d" % (
            transport or "default", str(r0["mode"]), same_sel,
            float(np.max(np.abs(r0["cost"] - cost) / np.abs(cost))), bool(np.array_equal(r0["X"], X)), int(r0["posts"])),
            flush=True)
