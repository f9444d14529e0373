"""scratch: replay time of the lattice agent / sphere2500 / tiers under DCORA_SP_LANES (read once per process)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, dcora_amd as da
from dcora_amd import datasets, synth
os.environ["DCORA_PRECOND"] = "sparse"
def one(case, r, d, n, Q, **kw):
    k = Q.n
    P = da.QuadraticProblem(r, d, n, Q, G=np.zeros((r, k)), reg=0.1, **kw)
    P.f(np.zeros((r, k)))
    ms, nbytes = P.time_precond(reps=200)
    print(json.dumps({"case": case, "lanes": os.environ.get("DCORA_SP_LANES", "default"), "us": round(1e3 * ms, 2), "launches": P.precond_info().get("launches")}), flush=True)
    P.close()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
lat = synth.lattice_se3()
nb, ids, vals = bench.agent_block(lat, 8, 0)
one("lattice100k/8", 5, 3, nb, da.build_Q_pgo(lat, n=nb, agent=0, ids=ids, vals=vals))
if which == "all":
    ds = datasets.product_dataset("sphere2500")
    one("sphere2500/1", 5, ds.d, ds.n, da.build_Q_pgo(ds))
    ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
    one("tiers", 2, ra.d, ra.n, ra.Q, l=ra.l, b=ra.b)
