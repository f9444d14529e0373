"""The smallest and the widest inputs the operator accepts (ref tests: the reference's fixtures stop at 6 poses; its
classes take any n >= 1, r >= d -- src/QuadraticProblem.cpp:19-34, src/manifold/LiftedManifold.cpp:18-35): one pose
without measurements, two poses and one measurement, the minimum rank r = d and the widest r = 16, agents of a single
pose.  HIP path through the C ABI against the CPU oracle on the same inputs."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import dcora_amd as da
    from oracle import orc
    if da.device_count() < 1:
        pytest.fail("no GPU visible: the product has no CPU fallback")
    return da, orc


def _tiny(env, n):
    """n = 1: no measurement at all (Q = 0); n = 2: the first measurement of tinyGrid3D between poses 0 and 1"""
    da, orc = env
    ds = common.product_dataset("tinyGrid3D")
    if n == 1:
        ids, vals = np.zeros((0, 4), np.int32), np.zeros((0, ds.vals.shape[1]))
    else:
        ids, vals = ds.ids[:1].copy(), ds.vals[:1].copy()
        ids[0] = [0, 0, 0, 1]
    return da.Dataset(3, n, ids, vals), orc.Dataset(3, n, ids, vals)


def _close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("r", [3, 5, 16])
def test_operators_on_the_smallest_problems(env, n, r):
    da, orc = env
    d1, do = _tiny(env, n)
    d = 3
    rng = np.random.default_rng(7)
    G = rng.standard_normal((r, (d + 1) * n))
    P = da.QuadraticProblem(r, d, n, da.build_Q_pgo(d1), G=G)
    Po = orc.Problem(r, d, n, orc.build_Q_pgo(do), G=G)
    X = common.random_point(r, d, n, 5, orc.project_to_manifold)
    V = orc.tangent_project(r, d, n, X, common.random_tangent(r, d, n, 6))
    assert abs(P.f(X) - Po.f(X)) <= 1e-12 * max(1.0, abs(Po.f(X)))
    assert _close(P.EucGrad(X), Po.egrad(X), 1e-13)
    assert _close(P.RieGrad(X), Po.rgrad(X), 1e-12)
    assert _close(P.HessVec(X, V), Po.hess(X, V), 1e-12)
    assert _close(P.Retract(X, 0.3 * V), orc.retract(r, d, n, X, 0.3 * V), 1e-13)
    assert _close(P.PreCondition(X, V), Po.precondition(X, V), 1e-10)
    assert _close(da.manifold_project(r, d, n, X + 0.2 * V), orc.project_to_manifold(r, d, n, X + 0.2 * V), 1e-12)
    P.close()


@pytest.mark.parametrize("r", [3, 16])
def test_rtr_on_two_poses_and_one_measurement(env, r):
    """f = 0 is attained (one relative pose can always be met): RTR gets there from a random start"""
    da, orc = env
    d1, do = _tiny(env, 2)
    P = da.QuadraticProblem(r, 3, 2, da.build_Q_pgo(d1))
    Po = orc.Problem(r, 3, 2, orc.build_Q_pgo(do))
    X0 = common.random_point(r, 3, 2, 9, orc.project_to_manifold)
    prm = dict(RTR_iterations=30, gradnorm_tol=1e-8)
    opt = da.QuadraticOptimizer(P, da.ROptParameters(**prm))
    X = opt.optimize(X0)
    res = opt.getOptResult()
    Xo, reso = Po.optimize(X0, **prm)
    assert res["fOpt"] < 1e-12 and reso["fOpt"] < 1e-12
    assert res["gradNormOpt"] < 1e-6
    assert abs(Po.f(X) - res["fOpt"]) < 1e-12
    P.close()


def test_rbcd_with_agents_of_a_single_pose(env):
    """tinyGrid3D split over as many agents as it has poses: every block is one pose, every measurement is shared"""
    da, orc = env
    ds, dso = common.product_dataset("tinyGrid3D"), common.oracle_dataset("tinyGrid3D")
    r, R, iters = 5, ds.n, 40
    X0 = common.random_point(r, ds.d, ds.n, 2, orc.project_to_manifold)
    tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r, max_iters=iters, staircase=0, rgrad_tol=1e-12)
    s = da.RbcdSession(ds, num_robots=R, r=r)
    s.set_X(X0)
    out = s.run(max_iters=iters, rgrad_tol=1e-12)
    assert out["iters"] == iters
    assert np.array_equal(out["selected"], tr["selected"])
    assert np.allclose(out["cost"], tr["cost"], rtol=1e-8)
    assert common.rel(s.get_X(), tr["X"]) < 1e-7
    s.close()


def test_operators_at_the_widest_rank_on_a_real_graph(env):
    da, orc = env
    r = 16
    ds, dso = common.product_dataset("smallGrid3D"), common.oracle_dataset("smallGrid3D")
    P = da.QuadraticProblem(r, ds.d, ds.n, da.build_Q_pgo(ds))
    Po = orc.Problem(r, ds.d, ds.n, orc.build_Q_pgo(dso))
    X = common.random_point(r, ds.d, ds.n, 5, orc.project_to_manifold)
    V = orc.tangent_project(r, ds.d, ds.n, X, common.random_tangent(r, ds.d, ds.n, 6))
    assert abs(P.f(X) - Po.f(X)) <= 1e-12 * abs(Po.f(X))
    assert common.rel(P.RieGrad(X), Po.rgrad(X)) < 1e-12
    assert common.rel(P.HessVec(X, V), Po.hess(X, V)) < 1e-12
    assert common.rel(P.PreCondition(X, V), Po.precondition(X, V)) < 1e-9
    Xs, res = None, None
    opt = da.QuadraticOptimizer(P)
    Xs = opt.optimize(X)
    res = opt.getOptResult()
    Xo, reso = Po.optimize(X)
    assert res["outer_iterations"] == reso["outer_iters"] and res["inner_iterations"] == reso["inner_iters"]
    assert abs(res["fOpt"] - reso["fOpt"]) <= 1e-9 * abs(reso["fOpt"])
    P.close()


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("l,b", [(0, 0), (7, 0), (0, 5), (7, 5)])
def test_range_aided_layout_in_its_four_presence_cases(env, d, l, b):
    """ref src/manifold/LiftedManifold.cpp:67-88: the product manifold St(r,d)^n x OB(r,l) x R^(r x n) x R^(r x b) is
    built in four ways, with and without unit-sphere variables and landmarks.  A random sparse positive semidefinite Q
    of the layout's size stands in for the data matrix: every operator against the oracle, at r = d and r = d + 3"""
    import scipy.sparse as sp
    da, orc = env
    n = 12
    k = d * n + l + n + b
    rng = np.random.default_rng(100 * d + 10 * l + b)
    A = sp.random(3 * k, k, density=4.0 / k, random_state=np.random.RandomState(d + l + b), format="csr")
    Q = sp.csr_matrix(A.T @ A + 1e-3 * sp.identity(k))
    Q.sort_indices()
    for r in (d, d + 3):
        X = orc.project_to_manifold(r, d, n, rng.standard_normal((r, k)), l=l, b=b)
        V = rng.standard_normal((r, k))
        P = da.QuadraticProblem(r, d, n, da.Csr.from_scipy(Q), reg=0.05, l=l, b=b)
        Po = orc.Problem(r, d, n, orc.CSR.from_scipy(Q), reg=0.05, l=l, b=b)
        assert np.isclose(P.f(X), Po.f(X), rtol=1e-12)
        assert common.rel(P.RieGrad(X), Po.rgrad(X)) < 1e-12
        Vt = orc.tangent_project(r, d, n, X, V, l=l, b=b)
        assert common.rel(P.projectToTangentSpace(X, V), Vt) < 1e-13
        assert common.rel(P.HessVec(X, Vt), Po.hess(X, Vt)) < 1e-11
        assert common.rel(P.Retract(X, 0.2 * Vt), orc.retract(r, d, n, X, 0.2 * Vt, l=l, b=b)) < 1e-13
        assert common.rel(P.PreCondition(X, Vt), Po.precondition(X, Vt)) < 1e-9
        M = X + 0.3 * V
        assert common.rel(da.manifold_project(r, d, n, M, l=l, b=b),
                          orc.project_to_manifold(r, d, n, M, l=l, b=b)) < 1e-12
        S = da.dual_certificate(r, d, n, X, da.Csr.from_scipy(Q), l=l, b=b).to_scipy()
        So = orc.dual_certificate(r, d, n, X, orc.CSR.from_scipy(Q), l=l, b=b).to_scipy()
        assert abs(S - So).max() < 1e-9 * max(1.0, abs(So).max())
        opt = da.QuadraticOptimizer(P)
        Xs = opt.optimize(X)
        res = opt.getOptResult()
        Xo, reso = Po.optimize(X)
        assert res["outer_iterations"] == reso["outer_iters"] and res["inner_iterations"] == reso["inner_iters"]
        assert abs(res["fOpt"] - reso["fOpt"]) <= 1e-9 * abs(reso["fOpt"])
        P.close()


@pytest.mark.parametrize("d", [2, 3])
def test_range_aided_ordering_without_ranges_or_landmarks(env, d):
    """The reference picks the manifold by GRAPH TYPE (ref src/Graph.cpp:68-75, src/QuadraticProblem.cpp:19-34): a
    RangeAidedSLAMGraph that holds only pose-pose measurements still orders its columns [R_1 .. R_n | t_1 .. t_n].
    dims.layout = DCORA_LAYOUT_RA carries that; every operator must equal the SE-ordered oracle on the permuted
    matrix (with l = b = 0 and layout AUTO the library would read the translation columns as rotation columns)."""
    import scipy.sparse as sp
    from dcora_amd import capi
    da, orc = env
    n = 12
    k = (d + 1) * n
    rng = np.random.default_rng(40 + d)
    A = sp.random(3 * k, k, density=4.0 / k, random_state=np.random.RandomState(d), format="csr")
    Qra = sp.csr_matrix(A.T @ A + 1e-3 * sp.identity(k))
    perm = np.array([i * (d + 1) + c for i in range(n) for c in range(d)] + [i * (d + 1) + d for i in range(n)])
    inv = np.argsort(perm)            # column of the RA ordering that sits at SE column j
    Qse = sp.csr_matrix(Qra[inv][:, inv])
    Qra.sort_indices()
    Qse.sort_indices()
    RA = capi.LAYOUT_RA
    for r in (d, d + 3):
        Xse = orc.project_to_manifold(r, d, n, rng.standard_normal((r, k)))
        Vse = rng.standard_normal((r, k))
        X, V = Xse[:, perm], Vse[:, perm]
        P = da.QuadraticProblem(r, d, n, da.Csr.from_scipy(Qra), reg=0.05, layout=RA)
        Po = orc.Problem(r, d, n, orc.CSR.from_scipy(Qse), reg=0.05)
        assert np.isclose(P.f(X), Po.f(Xse), rtol=1e-12)
        assert common.rel(P.RieGrad(X), Po.rgrad(Xse)[:, perm]) < 1e-12
        Vt = orc.tangent_project(r, d, n, Xse, Vse)
        assert common.rel(P.projectToTangentSpace(X, V), Vt[:, perm]) < 1e-13
        assert common.rel(P.HessVec(X, Vt[:, perm]), Po.hess(Xse, Vt)[:, perm]) < 1e-11
        assert common.rel(P.Retract(X, 0.2 * Vt[:, perm]), orc.retract(r, d, n, Xse, 0.2 * Vt)[:, perm]) < 1e-13
        assert common.rel(P.PreCondition(X, Vt[:, perm]), Po.precondition(Xse, Vt)[:, perm]) < 1e-9
        M = Xse + 0.3 * Vse
        assert common.rel(da.manifold_project(r, d, n, M[:, perm], layout=RA),
                          orc.project_to_manifold(r, d, n, M)[:, perm]) < 1e-12
        S = da.dual_certificate(r, d, n, X, da.Csr.from_scipy(Qra), layout=RA).to_scipy()
        So = orc.dual_certificate(r, d, n, Xse, orc.CSR.from_scipy(Qse)).to_scipy()
        assert abs(S - So[perm][:, perm]).max() < 1e-9 * max(1.0, abs(So).max())
        opt = da.QuadraticOptimizer(P)
        Xs = opt.optimize(X)
        res = opt.getOptResult()
        Xo, reso = Po.optimize(Xse)
        assert res["outer_iterations"] == reso["outer_iters"] and res["inner_iterations"] == reso["inner_iters"]
        assert abs(res["fOpt"] - reso["fOpt"]) <= 1e-9 * abs(reso["fOpt"])
        assert common.rel(Xs, Xo[:, perm]) < 1e-6
        # the same dims without the flag is the SE ordering: a different problem on this matrix
        Pse = da.QuadraticProblem(r, d, n, da.Csr.from_scipy(Qra), reg=0.05)
        assert abs(Pse.f(X) - P.f(X)) <= 1e-12 * abs(P.f(X)) and common.rel(Pse.RieGrad(X), P.RieGrad(X)) > 1e-3
        Pse.close()
        P.close()
    with pytest.raises(da.DcoraError):
        da.QuadraticProblem(d, d, n, da.Csr.from_scipy(sp.identity(k + 3, format="csr")), l=3, layout=capi.LAYOUT_SE)
