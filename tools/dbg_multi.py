import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common
import dcora_amd as da
name, R, r, iters = "sphere2500", 5, 5, 40
ds = common.product_dataset(name)
X0 = common.random_point(r, ds.d, ds.n, 11, lambda r_, d_, n_, M: da.manifold_project(r_, d_, n_, M))
s1 = da.RbcdSession(ds, num_robots=R, r=r)
s1.set_X(X0)
world = 2
ss = [da.RbcdSession(ds, num_robots=R, r=r, rank=k, world_size=world) for k in range(world)]
per = (R + world - 1) // world
owner = [a // per for a in range(R)]
dh = ds.d + 1
slot = r * dh * max(ss[0].public_count(a) for a in range(R))
buf = torch.zeros(slot, dtype=torch.float64, device="cuda")
evs = [torch.zeros(2 * R, dtype=torch.float64, device="cuda") for _ in range(world)]
for s in ss: s.set_X(X0)
def move(a):
    src = ss[owner[a]]
    src.pack_public_dev(a, buf.data_ptr()); src.synchronize()
    for k, s in enumerate(ss):
        if k != owner[a]:
            s.unpack_public_dev(a, buf.data_ptr()); s.synchronize()
def blocks(Xs):
    out = np.zeros_like(Xs[0])
    for a in range(R):
        n_a = ds.n // R if a < R - 1 else ds.n - (R - 1) * (ds.n // R)
        c0 = a * (ds.n // R) * dh
        out[:, c0:c0 + n_a * dh] = Xs[owner[a]][:, c0:c0 + n_a * dh]
    return out
selected = 0
for it in range(iters):
    s1.phase_nonselected(selected); s1.synchronize()
    for s in ss: s.phase_nonselected(selected); s.synchronize()
    Xa = s1.get_X(); Xb = blocks([s.get_X() for s in ss])
    dA = np.max(np.abs(Xa - Xb))
    for a in range(R):
        if a != selected: move(a)
    s1.phase_selected(selected); s1.synchronize()
    for s in ss: s.phase_selected(selected); s.synchronize()
    move(selected)
    Xa = s1.get_X(); Xb = blocks([s.get_X() for s in ss])
    dB = np.max(np.abs(Xa - Xb))
    res1 = s1.last_result(); res2 = ss[owner[selected]].last_result()
    c2, gn, bn, nxt = s1.evaluate()
    tot = torch.zeros(2 * R, dtype=torch.float64, device="cuda")
    for k, s in enumerate(ss):
        s.phase_evaluate_dev(evs[k].data_ptr()); s.synchronize(); tot += evs[k]
    h = tot.cpu().numpy()
    print(it, selected, "after nonselected %.3g after selected %.3g" % (dA, dB), res1["inner_iterations"], res2["inner_iterations"], res1["fOpt"] - res2["fOpt"], nxt, int(np.argmax(h[0::2])))
    selected = nxt
