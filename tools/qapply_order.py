"""experiment: the block Q-apply on the 100k lattice under a locality ordering of the poses (sub-cubes of the lattice
instead of the snake order of the trajectory): warm / cold us per launch.  python tools/qapply_order.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
from dcora_amd import synth

def coords_of(nx=50, ny=50, nz=40):
    n = nx * ny * nz
    c = np.empty((n, 3), np.int64)
    idx = 0
    for z in range(nz):
        ys = range(ny) if z % 2 == 0 else range(ny - 1, -1, -1)
        for yi, y in enumerate(ys):
            xs = range(nx) if yi % 2 == 0 else range(nx - 1, -1, -1)
            for x in xs:
                c[idx] = (x, y, z); idx += 1
    return c

def measure(tag, ds, r=5):
    Q = da.build_Q_pgo(ds)
    k = 4 * ds.n
    Ps = [da.QuadraticProblem(r, 3, ds.n, Q, G=np.zeros((r, k)), reg=-1.0) for _ in range(4)]
    for P in Ps: P.f(np.zeros((r, k)))
    ms, nb = Ps[0].time_qapply(reps=50)
    msc = da.time_qapply_rotating(Ps, reps=48)
    print(json.dumps({"order": tag, "warm_us": round(ms * 1e3, 2), "cold_us": round(msc * 1e3, 2), "MB": round(nb / 1e6, 1),
                      "cold_frac": round(nb / (msc * 1e-3) / 8e12, 3)}), flush=True)
    for P in Ps: P.close()

ds = synth.lattice_se3()
measure("snake (trajectory order)", ds)
c = coords_of()
for (bx, by, bz) in ((4, 4, 2), (8, 4, 1), (4, 2, 4), (2, 2, 8)):
    key = ((c[:, 2] // bz) * 1000 + (c[:, 1] // by)) * 1000 + (c[:, 0] // bx)
    sub = ((c[:, 2] % bz) * by + (c[:, 1] % by)) * bx + (c[:, 0] % bx)
    perm = np.lexsort((sub, key))          # new position -> old pose
    inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
    ids = ds.ids.copy()
    ids[:, 1], ids[:, 3] = inv[ds.ids[:, 1]], inv[ds.ids[:, 3]]
    ds2 = synth.Dataset(3, ds.n, ids, ds.vals)
    measure("sub-cubes %dx%dx%d" % (bx, by, bz), ds2)
