"""The sparse replay of one 100k-lattice agent (k = 50 000) and of sphere2500 / tiers as single problems: us per
application, launches and stored bytes.
  python tools/bench_sp.py [lattice] [sphere] [tiers]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, common
import dcora_amd as da
from dcora_amd import synth
os.environ["DCORA_PRECOND"] = "sparse"
what = sys.argv[1:] or ["lattice"]
tag = {}
if "lattice" in what:
    lat = synth.lattice_se3()
    nb, ids, vals = bench.agent_block(lat, 8, 0)
    Q = da.build_Q_pgo(lat, n=nb, agent=0, ids=ids, vals=vals)
    k = 4 * nb
    for r in (5, 7):
        P = da.QuadraticProblem(r, 3, nb, Q, G=np.zeros((r, k)), reg=0.1)
        P.f(np.zeros((r, k)))
        for _ in range(2):
            ms, nbytes = P.time_precond(reps=200)
        info = P.precond_info()
        print(json.dumps(dict(tag, case="lattice_agent", r=r, us=round(ms * 1e3, 2), MB=round(nbytes / 1e6, 1),
                              launches=info.get("launches"), nnzL=info.get("nnzL"))), flush=True)
        P.close()
if "sphere" in what:
    ds = common.product_dataset("sphere2500")
    Q = da.build_Q_pgo(ds)
    for r in (5,):
        P = da.QuadraticProblem(r, 3, ds.n, Q, reg=0.1)
        ms, nbytes = P.time_precond(reps=200)
        print(json.dumps(dict(tag, case="sphere2500", r=r, us=round(ms * 1e3, 2), MB=round(nbytes / 1e6, 1),
                              launches=P.precond_info().get("launches"))), flush=True)
        P.close()
if "tiers" in what:
    ra = da.RADataset(os.path.join(common.DATA, "tiers.pyfg.gz"))
    for r in (2, 3):
        P = da.QuadraticProblem(r, ra.d, ra.n, ra.Q, G=np.zeros((r, ra.Q.n)), reg=0.1, l=ra.l, b=ra.b)
        P.f(np.zeros((r, ra.Q.n)))
        ms, nbytes = P.time_precond(reps=200)
        print(json.dumps(dict(tag, case="tiers", r=r, us=round(ms * 1e3, 2), MB=round(nbytes / 1e6, 1),
                              launches=P.precond_info().get("launches"))), flush=True)
        P.close()
