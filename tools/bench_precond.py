"""scratch benchmark (not a test): preconditioner application, dense inverse vs partitioned sparse inverse"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import common  # noqa: E402
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402


def one(tag, ds, robots, agent, r, kinds):
    if robots > 1:
        nb, ids, vals = bench.agent_block(ds, robots, agent)
        Q = da.build_Q_pgo(ds, n=nb, agent=agent, ids=ids, vals=vals)
    else:
        nb, Q = ds.n, da.build_Q_pgo(ds)
    k = (ds.d + 1) * nb
    for kind in kinds:
        os.environ["DCORA_PRECOND"] = kind
        t0 = time.time()
        try:
            P = da.QuadraticProblem(r, ds.d, nb, Q, G=np.zeros((r, k)), reg=0.1)
        except Exception as e:
            print(json.dumps({"case": tag, "kind": kind, "error": str(e)}), flush=True)
            continue
        setup = time.time() - t0
        P.f(np.zeros((r, k)))
        ms, nbytes = P.time_precond(reps=100)
        info = P.precond_info()
        print(json.dumps({"case": tag, "k": k, "r": r, "kind": kind, "us": ms * 1e3, "GBps": nbytes / ms / 1e6,
                          "bytes": nbytes, "setup_s": setup, "info": info}), flush=True)
        P.close()
    os.environ.pop("DCORA_PRECOND", None)


if __name__ == "__main__":
    sp = common.product_dataset("sphere2500")
    one("sphere2500/5", sp, 5, 0, 5, ["dense", "sparse"])
    one("sphere2500/2", sp, 2, 0, 5, ["dense", "sparse"])
    one("sphere2500/1", sp, 1, 0, 5, ["dense", "sparse"])
    to = common.product_dataset("torus3D")
    one("torus3D/1", to, 1, 0, 5, ["dense", "sparse"])
    lat = synth.lattice_se3()
    one("lattice100k/8", lat, 8, 0, 5, ["sparse"])
    one("lattice100k/8 r7", lat, 8, 0, 7, ["sparse"])
    if "--big" in sys.argv:
        one("lattice100k/1", lat, 1, 0, 5, ["sparse"])
