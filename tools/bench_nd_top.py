"""sparse preconditioner application with / without the merged dense top of the dissection tree (DCORA_ND_TOP, read once
per process: run this script once per setting)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dcora_amd as da  # noqa: E402
from dcora_amd import datasets, synth  # noqa: E402


def one(case, r, d, n, Q, **kw):
    k = Q.n
    t0 = time.perf_counter()
    P = da.QuadraticProblem(r, d, n, Q, G=np.zeros((r, k)), reg=0.1, **kw)
    setup = time.perf_counter() - t0
    P.f(np.zeros((r, k)))
    ms, nbytes = P.time_precond(reps=200)
    print(json.dumps({"case": case, "top": os.environ.get("DCORA_ND_TOP", "default"), "k": k, "r": r, "us": 1e3 * ms,
                      "MB": nbytes / 1e6, "setup_s": setup, "info": P.precond_info()}), flush=True)
    P.close()


os.environ["DCORA_PRECOND"] = "sparse"
for name in ("sphere2500", "torus3D"):
    ds = datasets.product_dataset(name)
    one(name + "/1", 5, ds.d, ds.n, da.build_Q_pgo(ds))
ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
one("tiers", 2, ra.d, ra.n, ra.Q, l=ra.l, b=ra.b)
lat = synth.lattice_se3()
nb, ids, vals = bench.agent_block(lat, 8, 0)
one("lattice100k/8", 5, 3, nb, da.build_Q_pgo(lat, n=nb, agent=0, ids=ids, vals=vals))
