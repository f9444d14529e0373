"""Generates tests/golden/vectors.npz: seeded inputs and expected outputs of the hot path on the reference's small
datasets.  Expected values come from the CPU oracle (oracle/), each cross-checked here against an independent
numpy/scipy computation before it is written (the script aborts on any disagreement).  Re-run only when the oracle
changes on purpose:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spla

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import common  # noqa: E402
import g2o_np  # noqa: E402
from oracle import orc  # noqa: E402

out = {}
for name, r in (("tinyGrid3D", 4), ("smallGrid3D", 5), ("pose_graph_optimization_test_2d", 3)):
    ds = common.oracle_dataset(name)
    g = g2o_np.read_g2o(common.data_path(name))
    d, n = ds.d, ds.n
    dh = d + 1
    Q = orc.build_Q_pgo(ds)
    Qd = g2o_np.dense_Q(g)
    assert np.abs(Q.to_scipy().toarray() - Qd).max() < 1e-10 * np.abs(Qd).max()
    rng = np.random.default_rng(1234)
    G = rng.standard_normal((r, dh * n))
    X = orc.project_to_manifold(r, d, n, rng.uniform(-1, 1, (r, dh * n)))
    V = rng.standard_normal((r, dh * n))
    P = orc.Problem(r, d, n, Q, G=G)
    f = P.f(X)
    assert np.isclose(f, 0.5 * np.sum((X @ Qd) * X) + np.sum(X * G), rtol=1e-12)
    eg = P.egrad(X)
    assert np.allclose(eg, X @ Qd + G, rtol=1e-11, atol=1e-9)
    rg = P.rgrad(X)
    Vt = orc.tangent_project(r, d, n, X, V)
    for i in range(n):
        Y, Vi = X[:, dh * i:dh * i + d], V[:, dh * i:dh * i + d]
        YtV = Y.T @ Vi
        assert np.allclose(Vt[:, dh * i:dh * i + d], Vi - Y @ (0.5 * (YtV + YtV.T)), atol=1e-13)
    Hv = P.hess(X, Vt)
    Z = P.precondition(X, Vt)
    M = (Q.to_scipy() + 0.1 * sp.identity(Q.n)).tocsc()
    Zs = orc.tangent_project(r, d, n, X, spla.splu(M).solve(Vt.T).T)
    assert np.linalg.norm(Z - Zs) < 1e-9 * np.linalg.norm(Zs)
    Rt = orc.retract(r, d, n, X, 0.3 * Vt)
    Mm = X + 0.2 * V
    Pm = orc.project_to_manifold(r, d, n, Mm)
    for i in range(n):
        Qf, Rf = np.linalg.qr((X + 0.3 * Vt)[:, dh * i:dh * i + d])
        assert np.allclose(Rt[:, dh * i:dh * i + d], Qf * np.sign(np.diag(Rf)), atol=1e-12)
        U, _ = scipy.linalg.polar(Mm[:, dh * i:dh * i + d])
        assert np.allclose(Pm[:, dh * i:dh * i + d], U, atol=1e-12)
    S = orc.dual_certificate(r, d, n, X, Q).to_scipy().toarray()
    lam_min = np.linalg.eigvalsh(S)[0]
    Xo, res = P.optimize(X)  # reference defaults: RTR 3 x 50, tol 1e-2
    key = name + "/"
    out.update({key + "r": r, key + "G": G, key + "X": X, key + "V": V, key + "f": f, key + "egrad": eg,
                key + "rgrad": rg, key + "Vt": Vt, key + "hess": Hv, key + "precond": Z, key + "retract": Rt,
                key + "M": Mm, key + "polar": Pm, key + "lambda_min_S": lam_min, key + "rtr_fOpt": res["fOpt"],
                key + "rtr_outer": res["outer_iters"], key + "rtr_inner": res["inner_iters"], key + "rtr_X": Xo})

# RBCD++ trace, 5 agents, smallGrid3D, rank 5
ds = common.oracle_dataset("smallGrid3D")
rng = np.random.default_rng(77)
X0 = orc.project_to_manifold(5, 3, ds.n, rng.uniform(-1, 1, (5, 4 * ds.n)))
tr = orc.run_rbcd(ds, X0, num_robots=5, r_min=5, max_iters=40, staircase=0, rgrad_tol=0.0)
out.update({"rbcd/X0": X0, "rbcd/cost": tr["cost"], "rbcd/gradnorm": tr["gradnorm"], "rbcd/selected": tr["selected"]})
# converged + certified run (published optimum of smallGrid3D: 2 f = 1025.398)
tr = orc.run_rbcd(ds, X0, num_robots=5, r_min=5, max_iters=1000, staircase=1)
assert tr["certified"] == 1 and abs(tr["cost"][-1] - 1025.398) < 1e-2
out.update({"rbcd/final_cost": tr["cost"][-1], "rbcd/final_iters": tr["total_iters"]})
np.savez_compressed(os.path.join(HERE, "vectors.npz"), **out)
print("wrote", os.path.join(HERE, "vectors.npz"), len(out), "arrays")
