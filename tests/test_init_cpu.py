"""chordalInitialization (ref src/DCORA_solver.cpp:218-268): oracle and product host code against an independent
least-squares solve of the reference's B matrices (ref src/DCORA_utils.cpp:1542-1630) with scipy."""
import numpy as np
import pytest
import scipy.sparse as sp

import common
import g2o_np


def chordal_scipy(g):
    d, n, edges = g["d"], g["n"], g["edges"]
    m, d2 = len(edges), g["d"] ** 2
    # B3 (eq. 69c): rows e*d2 + d*r + l, columns i*d2 + d*c + l  with  -sqrt(kappa) R(c, r);  + sqrt(kappa) I at j
    I, J, V = [], [], []
    for e, (i, j, R, t, kappa, tau) in enumerate(edges):
        sk = np.sqrt(kappa)
        for r in range(d):
            for c in range(d):
                for l in range(d):
                    I.append(e * d2 + d * r + l); J.append(i * d2 + d * c + l); V.append(-sk * R[c, r])
        for l in range(d2):
            I.append(e * d2 + l); J.append(j * d2 + l); V.append(sk)
    B3 = sp.csr_matrix((V, (I, J)), shape=(d2 * m, d2 * n))
    cR = B3[:, :d2] @ np.eye(d).reshape(-1, order="F")
    rvec = -np.linalg.lstsq(B3[:, d2:].toarray(), cR, rcond=None)[0]
    Rch = np.zeros((d, d * n))
    Rch[:, :d] = np.eye(d)
    Rch[:, d:] = rvec.reshape((d, d * (n - 1)), order="F")
    for i in range(1, n):
        U, _, Vt = np.linalg.svd(Rch[:, d * i:d * i + d])
        if np.linalg.det(U) * np.linalg.det(Vt) < 0:
            U[:, -1] *= -1
        Rch[:, d * i:d * i + d] = U @ Vt
    # B1 / B2 (eq. 69a, 69b)
    I, J, V = [], [], []
    for e, (i, j, R, t, kappa, tau) in enumerate(edges):
        st = np.sqrt(tau)
        for l in range(d):
            I += [e * d + l, e * d + l]; J += [i * d + l, j * d + l]; V += [-st, st]
    B1 = sp.csr_matrix((V, (I, J)), shape=(d * m, d * n))
    I, J, V = [], [], []
    for e, (i, j, R, t, kappa, tau) in enumerate(edges):
        st = np.sqrt(tau)
        for k in range(d):
            for r in range(d):
                I.append(d * e + r); J.append(d2 * i + d * k + r); V.append(-st * t[k])
    B2 = sp.csr_matrix((V, (I, J)), shape=(d * m, d2 * n))
    c = B2 @ Rch.reshape(-1, order="F")
    tred = -np.linalg.lstsq(B1[:, d:].toarray(), c, rcond=None)[0]
    T = np.zeros((d, (d + 1) * n))
    for i in range(n):
        T[:, (d + 1) * i:(d + 1) * i + d] = Rch[:, d * i:d * i + d]
        if i > 0:
            T[:, (d + 1) * i + d] = tred[d * (i - 1):d * i]
    return T


@pytest.mark.parametrize("name", ["pose_graph_optimization_test_2d", "pose_graph_optimization_test_3d", "tinyGrid3D",
                                  "smallGrid3D"])
def test_chordal_initialization(built, name):
    import dcora_amd as da
    from oracle import orc
    g = g2o_np.read_g2o(common.data_path(name))
    Ts = chordal_scipy(g)
    To = orc.chordal_initialization(common.oracle_dataset(name))
    Tp = da.chordal_initialization(common.product_dataset(name))
    assert np.abs(To - Ts).max() < 1e-7
    assert np.abs(Tp - Ts).max() < 1e-7
    assert np.abs(Tp - To).max() < 1e-10
    if name.startswith("pose_graph"):
        # noiseless fixtures: chordal initialisation recovers the ground truth in the frame of pose 0
        X = g2o_np.ground_truth_X(g)
        d = g["d"]
        R0, t0 = X[:, :d], X[:, d]
        for i in range(g["n"]):
            Ri, ti = X[:, (d + 1) * i:(d + 1) * i + d], X[:, (d + 1) * i + d]
            assert np.allclose(Tp[:, (d + 1) * i:(d + 1) * i + d], R0.T @ Ri, atol=1e-8)
            assert np.allclose(Tp[:, (d + 1) * i + d], R0.T @ (ti - t0), atol=1e-8)


def test_chordal_then_rbcd_reaches_the_published_optimum_of_sphere2500(built):
    """the oracle's full flow from the chordal start: 5 agents, certified at rank 5, 2 f = 1687.0 (SE-Sync's value)"""
    from oracle import orc
    ds = common.oracle_dataset("sphere2500")
    T = orc.chordal_initialization(ds)
    X0 = np.zeros((5, 4 * ds.n))
    X0[:3] = T
    tr = orc.run_rbcd(ds, X0, num_robots=5, max_iters=1000)
    assert tr["certified"] == 1 and tr["final_rank"] == 5
    assert abs(tr["cost"][-1] - 1687.02) < 0.05
    assert tr["gradnorm"][-1] < 0.1
