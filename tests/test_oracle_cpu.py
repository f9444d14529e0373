"""CPU tests (no GPU): pin the oracle against the reference's own known-answer fixtures and against independent
numpy/scipy computations, check the product's host-side data feed against both, and check that the C-ABI library
loads and exports every symbol include/dcora_hip.h declares."""
import os
import re

import numpy as np
import pytest
import scipy.linalg
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import common
import g2o_np


@pytest.fixture(scope="module")
def orc(built):
    from oracle import orc as o
    return o


@pytest.fixture(scope="module")
def da(built):
    import dcora_amd
    return dcora_amd


ALL = ["pose_graph_optimization_test_2d", "pose_graph_optimization_test_3d", "tinyGrid3D", "smallGrid3D"]


# ---- data feed -----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ALL + ["sphere2500"])
def test_readers_and_Q_against_numpy(orc, da, name):
    g = g2o_np.read_g2o(common.data_path(name))
    dso, dsp = common.oracle_dataset(name), common.product_dataset(name)
    assert (dso.d, dso.n, dso.m) == (g["d"], g["n"], len(g["edges"])) == (dsp.d, dsp.n, dsp.m)
    d = g["d"]
    for ds in (dso, dsp):
        for k in (0, len(g["edges"]) // 2, len(g["edges"]) - 1):
            i, j, R, t, kappa, tau = g["edges"][k]
            assert (ds.ids[k, 1], ds.ids[k, 3]) == (i, j)
            assert np.allclose(ds.vals[k, :d * d].reshape(d, d).T, R, atol=1e-14)
            assert np.allclose(ds.vals[k, d * d:d * d + d], t, atol=0)
            assert np.isclose(ds.vals[k, d * d + d], kappa, rtol=1e-13)
            assert np.isclose(ds.vals[k, d * d + d + 1], tau, rtol=1e-13)
    Qo, Qp = orc.build_Q_pgo(dso).to_scipy(), da.build_Q_pgo(dsp).to_scipy()
    assert abs(Qo - Qp).max() < 1e-11 * abs(Qo).max()
    if g["n"] <= 200:
        Qd = g2o_np.dense_Q(g)
        assert np.abs(Qo.toarray() - Qd).max() < 1e-11 * np.abs(Qd).max()
    # symmetric, and the translation gauge is in the null space: Q * [0..0 1]^T blocks = 0
    assert abs(Qp - Qp.T).max() < 1e-12 * abs(Qp).max()
    ones = np.zeros(Qp.shape[0])
    ones[d::d + 1] = 1.0
    assert np.abs(Qp @ ones).max() < 1e-9 * abs(Qp).max()


def test_nnz_counts_match_survey(da):
    # measured nnz of the reference pattern
    assert da.build_Q_pgo(common.product_dataset("sphere2500")).nnz == 168662
    assert da.build_Q_pgo(common.product_dataset("smallGrid3D")).nnz == 9722


@pytest.mark.parametrize("name", ALL)
def test_cost_matches_edgewise_form(orc, name):
    g = g2o_np.read_g2o(common.data_path(name))
    ds = common.oracle_dataset(name)
    r = ds.d + 2
    X = common.random_point(r, ds.d, ds.n, 0, orc.project_to_manifold)
    P = orc.Problem(r, ds.d, ds.n, orc.build_Q_pgo(ds))
    assert np.isclose(P.f(X), g2o_np.edgewise_cost(g, X), rtol=1e-12)
    # gradient against central finite differences of the edge-wise cost along a random direction
    V = common.random_tangent(r, ds.d, ds.n, 1)
    h = 1e-6
    fd = (g2o_np.edgewise_cost(g, X + h * V) - g2o_np.edgewise_cost(g, X - h * V)) / (2 * h)
    assert np.isclose(np.sum(P.egrad(X) * V), fd, rtol=1e-6)


# ---- reference fixtures ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["pose_graph_optimization_test_2d", "pose_graph_optimization_test_3d"])
def test_noiseless_dataset_known_answers(orc, name):
    """ref tests/testAgent.cpp:23-28, 101-155: the VERTEX lines are the optimum: cost 0, zero gradient, fixed point of
    the local solver; hence Lambda = 0, S = Q is PSD (certified)"""
    g = g2o_np.read_g2o(common.data_path(name))
    ds = common.oracle_dataset(name)
    X = g2o_np.ground_truth_X(g)
    Q = orc.build_Q_pgo(ds)
    P = orc.Problem(ds.d, ds.d, ds.n, Q)
    assert abs(P.f(X)) < 1e-15
    assert np.linalg.norm(P.rgrad(X)) < 1e-7
    Xn, res = P.optimize(X)
    assert np.abs(Xn - X).max() < 1e-9
    S = orc.dual_certificate(ds.d, ds.d, ds.n, X, Q)
    assert abs(S.to_scipy() - Q.to_scipy()).max() < 1e-7
    assert orc.fast_verification(S, 1e-3, block=ds.d + 1)[0]
    # one full accelerated RBCD round with 2 agents keeps the ground truth (ref tests/testAgent.cpp:290-456)
    tr = orc.run_rbcd(ds, X, num_robots=2, r_min=ds.d, max_iters=4, staircase=0, rgrad_tol=0.0)
    assert np.abs(tr["X"] - X).max() < 1e-9


def test_prior_known_answer(orc):
    """ref tests/testRobust.cpp:162-226 (testPrior): 2 poses, one odometry edge (R = I, t = 0, kappa 1e4, tau 100),
    prior on pose 1 with kappa 1e4 / tau 100 entering through G = -P Omega (ref src/Graph.cpp:805-816); RTR
    (50 x 500, tol 1e-5) from the odometry initialisation converges to the prior within 1e-6."""
    d, n, r = 3, 2, 3
    ids = np.array([[0, 0, 0, 1]], np.int32)
    vals = np.concatenate([np.eye(3).T.reshape(-1), np.zeros(3), [10000.0, 100.0, 1.0]])[None, :]
    ds = orc.Dataset(d, n, ids, vals)
    Q = orc.build_Q_pgo(ds)
    A = np.array([[0.7236, 0.1817, 0.6658], [-0.6100, 0.6198, 0.4938], [-0.3230, -0.7634, 0.5594]])
    Rp = orc.project_to_rotation_group(A)
    assert np.allclose(Rp.T @ Rp, np.eye(3), atol=1e-12) and np.linalg.det(Rp) > 0
    U, _, Vt = np.linalg.svd(A)
    assert np.allclose(Rp, U @ Vt, atol=1e-12)
    prior = np.zeros((3, 4))
    prior[:, :3] = Rp
    G = np.zeros((r, 4 * n))
    G[:, 4:8] = -prior @ np.diag([10000.0, 10000.0, 10000.0, 100.0])
    T = np.zeros((3, 8))
    T[:, 0:3] = np.eye(3)
    T[:, 4:7] = np.eye(3)  # odometryInitialization of (R = I, t = 0)
    P = orc.Problem(r, d, n, Q, G=G)
    assert np.linalg.norm(T[:, 0:4] - prior) > 1e-6
    Topt, res = P.optimize(T, RTR_iterations=50, RTR_tCG_iterations=500, gradnorm_tol=1e-5)
    assert np.linalg.norm(Topt[:, 0:4] - prior) < 1e-6
    assert np.linalg.norm(Topt[:, 4:8] - prior) < 1e-6


def test_tangent_projection_closed_form(orc):
    """ref tests/testManifold.cpp:354-390: Stiefel part = V - Y sym(Y^T V) (src/DCORA_utils.cpp:2033-2041),
    translations untouched"""
    r, d, n = 5, 3, 7
    X = common.random_point(r, d, n, 2, orc.project_to_manifold)
    V = common.random_tangent(r, d, n, 3)
    out = orc.tangent_project(r, d, n, X, V)
    for i in range(n):
        Y, Vi = X[:, 4 * i:4 * i + 3], V[:, 4 * i:4 * i + 3]
        YtV = Y.T @ Vi
        assert np.allclose(out[:, 4 * i:4 * i + 3], Vi - Y @ (0.5 * (YtV + YtV.T)), atol=1e-14)
        assert np.array_equal(out[:, 4 * i + 3], V[:, 4 * i + 3])
        # tangency: Y^T W is skew
        W = out[:, 4 * i:4 * i + 3]
        assert np.abs(Y.T @ W + W.T @ Y).max() < 1e-13


def test_ra_layout_projection_and_retraction(orc):
    """RA ordering [Y1..Yn | s1..sl | p1..pn | L1..Lb] (ref src/manifold/LiftedVariable.cpp:257-295):
    Stiefel blocks at i*d, unit spheres at d*n + i, the rest Euclidean (ref tests/testManifold.cpp:392-440)"""
    r, d, n, l, b = 4, 2, 5, 3, 2
    k = (d + 1) * n + l + b
    rng = np.random.default_rng(0)
    X = orc.project_to_manifold(r, d, n, rng.standard_normal((r, k)), l=l, b=b)
    for i in range(n):
        Y = X[:, d * i:d * i + d]
        assert np.abs(Y.T @ Y - np.eye(d)).max() < 1e-13
    assert np.allclose(np.linalg.norm(X[:, d * n:d * n + l], axis=0), 1.0, atol=1e-14)
    V = rng.standard_normal((r, k))
    W = orc.tangent_project(r, d, n, X, V, l=l, b=b)
    for i in range(l):
        y, v = X[:, d * n + i], V[:, d * n + i]
        assert np.allclose(W[:, d * n + i], v - y * (y @ v), atol=1e-14)  # src/DCORA_utils.cpp:2043-2051
    assert np.array_equal(W[:, d * n + l:], V[:, d * n + l:])
    Z = orc.retract(r, d, n, X, 0.1 * W, l=l, b=b)
    assert np.allclose(np.linalg.norm(Z[:, d * n:d * n + l], axis=0), 1.0, atol=1e-14)
    assert np.allclose(Z[:, d * n + l:], X[:, d * n + l:] + 0.1 * W[:, d * n + l:], atol=1e-15)


# ---- manifold arithmetic vs numpy/scipy ---------------------------------------------------------------------------
def test_polar_and_qf_against_lapack(orc):
    r, d, n = 6, 3, 9
    rng = np.random.default_rng(5)
    M = rng.standard_normal((r, 4 * n))
    P = orc.project_to_manifold(r, d, n, M)
    X = P
    V = 0.5 * rng.standard_normal((r, 4 * n))
    Z = orc.retract(r, d, n, X, V)
    for i in range(n):
        U, _ = scipy.linalg.polar(M[:, 4 * i:4 * i + 3])
        assert np.allclose(P[:, 4 * i:4 * i + 3], U, atol=1e-12)  # thin SVD -> U V^T (DCORA_utils.cpp:1677-1683)
        assert np.array_equal(P[:, 4 * i + 3], M[:, 4 * i + 3])
        Qf, Rf = np.linalg.qr((X + V)[:, 4 * i:4 * i + 3])
        Qf = Qf * np.sign(np.diag(Rf))  # qf: R with positive diagonal
        assert np.allclose(Z[:, 4 * i:4 * i + 3], Qf, atol=1e-12)
        assert np.allclose(Z[:, 4 * i + 3], (X + V)[:, 4 * i + 3], atol=1e-15)


def test_hessian_is_derivative_of_gradient(orc):
    """Riemannian Hessian (ROPTLIB EucHvToHv form) = covariant derivative of the Riemannian gradient along a
    retraction curve, projected: finite-difference check; also symmetric on the tangent space"""
    ds = common.oracle_dataset("tinyGrid3D")
    r = 5
    rng = np.random.default_rng(3)
    G = rng.standard_normal((r, 4 * ds.n))
    P = orc.Problem(r, ds.d, ds.n, orc.build_Q_pgo(ds), G=G)
    X = common.random_point(r, ds.d, ds.n, 4, orc.project_to_manifold)
    V = orc.tangent_project(r, ds.d, ds.n, X, common.random_tangent(r, ds.d, ds.n, 5))
    W = orc.tangent_project(r, ds.d, ds.n, X, common.random_tangent(r, ds.d, ds.n, 6))
    h = 1e-5
    gp = P.rgrad(orc.retract(r, ds.d, ds.n, X, h * V))
    gm = P.rgrad(orc.retract(r, ds.d, ds.n, X, -h * V))
    fd = orc.tangent_project(r, ds.d, ds.n, X, (gp - gm) / (2 * h))
    H = P.hess(X, V)
    assert common.rel(H, fd) < 1e-6
    assert np.isclose(np.sum(P.hess(X, V) * W), np.sum(P.hess(X, W) * V), rtol=1e-10)


# ---- sparse Cholesky / preconditioner / PSD ------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tinyGrid3D", "smallGrid3D", "sphere2500"])
def test_preconditioner_solve_against_scipy(orc, name):
    ds = common.oracle_dataset(name)
    r = 5
    Q = orc.build_Q_pgo(ds)
    P = orc.Problem(r, ds.d, ds.n, Q)
    V = common.random_tangent(r, ds.d, ds.n, 7)
    Z = P.precon_solve(V)
    M = (Q.to_scipy() + 0.1 * sp.identity(Q.n)).tocsc()
    Zs = spla.splu(M).solve(V.T).T
    assert common.rel(Z, Zs) < 1e-10


def test_product_host_psd_test(orc, da):
    ds = common.product_dataset("smallGrid3D")
    Q = da.build_Q_pgo(ds).to_scipy()
    for shift, expect in ((0.1, True), (1e-6, True), (-1e-3, False), (-1.0, False)):
        M = da.Csr.from_scipy(Q + shift * sp.identity(Q.shape[0]))
        assert da.is_psd(M, ds.d + 1) is expect
        assert orc.is_psd(orc.CSR.from_scipy(Q + shift * sp.identity(Q.shape[0])), ds.d + 1) is expect


# ---- certification ------------------------------------------------------------------------------------------------
def test_dual_certificate_and_min_eig_against_dense(orc):
    ds = common.oracle_dataset("smallGrid3D")
    r, d, n = 5, ds.d, ds.n
    Q = orc.build_Q_pgo(ds)
    X = common.random_point(r, d, n, 4, orc.project_to_manifold)
    S = orc.dual_certificate(r, d, n, X, Q).to_scipy().toarray()
    Qd = Q.to_scipy().toarray()
    QXt = Qd @ X.T
    Lam = np.zeros_like(Qd)
    for i in range(n):  # ref src/DCORA_utils.cpp:1908-1915
        Pm = QXt[4 * i:4 * i + 3, :] @ X[:, 4 * i:4 * i + 3]
        Lam[4 * i:4 * i + 3, 4 * i:4 * i + 3] = 0.5 * (Pm + Pm.T)
    assert np.abs(S - (Qd - Lam)).max() < 1e-10
    w = np.linalg.eigvalsh(S)
    ok, lam, v, mv = orc.min_eig(orc.CSR.from_scipy(sp.csr_matrix(S)), tol=1e-3)
    assert ok and abs(lam - w[0]) < 2e-3 * abs(w[0])
    assert abs(v @ S @ v - lam) < 1e-8 * abs(lam)
    ok, lam_lm, _, _ = orc.lanczos_lm(orc.CSR.from_scipy(sp.csr_matrix(S)), tol=1e-8)
    assert ok and np.isclose(abs(lam_lm), max(abs(w[0]), abs(w[-1])), rtol=1e-6)
    # critical point => S X^T = 0 (first-order condition), and the certificate passes at the global optimum
    P = orc.Problem(r, d, n, Q)
    Xo, res = P.optimize(X, RTR_iterations=300, RTR_tCG_iterations=300, gradnorm_tol=1e-9)
    So = orc.dual_certificate(r, d, n, Xo, Q)
    assert np.abs(So.to_scipy() @ Xo.T).max() < 1e-6
    assert np.isclose(2 * res["fOpt"], 1025.3980, rtol=1e-6)  # SE-Sync's published optimum of smallGrid3D
    assert orc.fast_verification(So, 1e-3, block=d + 1)[0]


def test_staircase_escape_decreases_cost(orc):
    ds = common.oracle_dataset("tinyGrid3D")
    r, d, n = 3, ds.d, ds.n
    Q = orc.build_Q_pgo(ds)
    P = orc.Problem(r, d, n, Q)
    rng = np.random.default_rng(8)
    # a rank-3 critical point reached from a random start may be a saddle of the rank-4 problem
    X, res = P.optimize(common.random_point(r, d, n, 21, orc.project_to_manifold), RTR_iterations=300,
                        RTR_tCG_iterations=300, gradnorm_tol=1e-9)
    S = orc.dual_certificate(r, d, n, X, Q)
    psd, theta, v, lmin = orc.fast_verification(S, 1e-3, block=d + 1)
    if not psd:
        Pn = orc.Problem(r + 1, d, n, Q)
        Xn = Pn.escape_saddle(X, theta, v)
        assert Xn is not None and Pn.f(Xn) < P.f(X)


# ---- C-ABI library -------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(da):
    from dcora_amd import capi
    hdr = open(capi.HEADER_PATH).read()
    declared = set(re.findall(r"\b(dcora_[A-Za-z0-9_]+)\s*\(", hdr))
    declared -= {"dcora_status"}
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    assert capi.lib().dcora_status_string(2).decode().startswith("no HIP device")


def test_no_cpu_fallback(da):
    """without a GPU every compute entry point must fail loudly (DCORA_ERR_NO_DEVICE)"""
    if da.device_count() > 0:
        pytest.skip("GPU present")
    ds = common.product_dataset("tinyGrid3D")
    with pytest.raises(da.DcoraError) as e:
        da.QuadraticProblem(3, ds.d, ds.n, da.build_Q_pgo(ds))
    assert e.value.status == 2
    with pytest.raises(da.DcoraError):
        da.RbcdSession(ds, num_robots=2, r=3)
    with pytest.raises(da.DcoraError):
        da.manifold_project(3, ds.d, ds.n, np.zeros((3, 4 * ds.n)))


def test_a_graph_without_measurements_builds(built):
    """one pose, no measurement: Q = 0 (k = d + 1), on the host builder and in the oracle"""
    import dcora_amd as da
    from oracle import orc
    ids, vals = np.zeros((0, 4), np.int32), np.zeros((0, 15))
    Q = da.build_Q_pgo(da.Dataset(3, 1, ids, vals)).to_scipy()
    Qo = orc.build_Q_pgo(orc.Dataset(3, 1, ids, vals)).to_scipy()
    assert Q.shape == (4, 4) and Qo.shape == (4, 4)
    assert abs(Q).sum() == 0 and abs(Qo).sum() == 0
