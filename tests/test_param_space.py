"""The RBCD session across its parameter space against the oracle: number of agents (1 ... more than a few poses
each), relaxation rank (d ... 8), planar and spatial graphs, acceleration on / off, RGD local solver."""
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


CASES = [
    # dataset, agents, rank, acceleration, iterations
    ("smallGrid3D", 1, 5, True, 6),
    ("smallGrid3D", 2, 3, True, 12),
    ("smallGrid3D", 7, 4, True, 20),
    ("smallGrid3D", 5, 8, True, 12),
    ("smallGrid3D", 5, 9, True, 35),        # beyond the fused kernels' rank limit: general kernels, crosses a restart
    ("smallGrid3D", 3, 12, False, 10),
    ("smallGrid3D", 5, 5, False, 15),
    ("smallGrid3D", 25, 5, True, 40),       # 5 poses per agent, crosses a restart
    ("tinyGrid3D", 3, 5, True, 10),
    ("pose_graph_optimization_test_2d", 2, 3, True, 8),
    ("pose_graph_optimization_test_3d", 2, 4, False, 8),
    ("CSAIL", 10, 2, True, 25),             # rank = d
    ("CSAIL", 3, 5, False, 10),
]


@pytest.mark.parametrize("name,R,r,accel,iters", CASES)
def test_rbcd_trace_matches_oracle(env, name, R, r, accel, iters):
    da, orc = env
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 17, orc.project_to_manifold)
    tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r, max_iters=iters, staircase=0, rgrad_tol=1e-12,
                      acceleration=int(accel))
    s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=accel)
    s.set_X(X0)
    out = s.run(max_iters=iters, rgrad_tol=1e-12)
    n = min(out["iters"], int(tr["total_iters"]))
    assert n == iters
    assert np.array_equal(out["selected"][:n], tr["selected"][:n])
    assert np.allclose(out["cost"][:n], tr["cost"][:n], rtol=1e-7)
    assert np.allclose(out["gradnorm"][:n], tr["gradnorm"][:n], rtol=1e-4, atol=1e-7)
    assert common.rel(s.get_X(), tr["X"]) < 1e-5


@pytest.mark.parametrize("name,R,r", [("smallGrid3D", 5, 5), ("CSAIL", 4, 3)])
def test_rbcd_with_rgd_local_solver(env, name, R, r):
    """ROptMethod::RGD as the agents' local solver (ref src/QuadraticOptimizer.cpp:123-150)"""
    da, orc = env
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 9, orc.project_to_manifold)
    iters = 12
    tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r, max_iters=iters, staircase=0, rgrad_tol=1e-12, method=1,
                      RGD_stepsize=1e-3, RGD_use_precond=1)
    prm = da.ROptParameters(method=da.ROptParameters.RGD, RGD_stepsize=1e-3, RGD_use_preconditioner=1)
    s = da.RbcdSession(ds, num_robots=R, r=r, params=prm)
    s.set_X(X0)
    out = s.run(max_iters=iters, rgrad_tol=1e-12)
    assert np.array_equal(out["selected"], tr["selected"])
    assert np.allclose(out["cost"], tr["cost"], rtol=1e-9)


@pytest.mark.parametrize("name,R,r_min,seed", [("tinyGrid3D", 3, 3, 21), ("smallGrid3D", 5, 3, 4)])
def test_riemannian_staircase_of_the_multi_robot_driver(env, name, R, r_min, seed):
    """examples/MultiRobotExample.cpp:172-372 from a random start at rank d: RBCD++ to a critical point, certificate,
    escape into the next rank, until fastVerification accepts.  The escape direction is an eigenvector (sign and
    Lanczos rounding are not canonical), so the two sides are compared on what the flow certifies: the optimum."""
    da, orc = env
    from dcora_amd import driver
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    X0 = common.random_point(r_min, ds.d, ds.n, seed, orc.project_to_manifold)
    kw = dict(max_iters=400, rgrad_tol=0.05)
    out = driver.multi_robot_example(ds, X0, num_robots=R, r_min=r_min, r_max=r_min + 6, **kw)
    tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r_min, r_max=r_min + 6, max_iters=400, rgrad_tol=0.05, staircase=1)
    assert out["certified"] and tr["certified"] == 1
    assert out["rank"] > r_min  # the staircase did climb
    assert [lv["escaped"] for lv in out["levels"][:-1]] == [True] * (len(out["levels"]) - 1)
    # the first level has no eigenvector in it: identical trace
    n0 = out["levels"][0]["iterations"]
    assert np.array_equal(out["selected"][:n0], tr["selected"][:n0])
    assert np.allclose(out["cost"][:n0], tr["cost"][:n0], rtol=1e-7)
    f_gpu, f_cpu = out["levels"][-1]["cost_2f"], tr["cost"][-1]
    assert abs(f_gpu - f_cpu) < 1e-3 * abs(f_cpu)  # levels end at |rgrad| < 0.05 or 400 iterations: same optimum
    # the oracle certifies the device's final point as well
    Qo = orc.build_Q_pgo(dso)
    So = orc.dual_certificate(out["rank"], ds.d, ds.n, out["X"], Qo)
    assert orc.fast_verification(So, 1e-3, block=ds.d + 1)[0]


def test_projection_to_the_manifold_is_feasible(env):
    """tests/testUtils.cpp:38-48, 82-134 (testStiefelProjection, testObliqueProjection, testProjectToSEMatrix,
    testProjectToRAMatrix) on the device kernels: d = 3, r = 5, n = 100 (l = 5, b = 7 for the RA layout), 50 draws
    for the single-block cases"""
    da, orc = env
    rng = np.random.default_rng(0)
    d, r, n, l, b = 3, 5, 100, 5, 7
    I = np.eye(d)
    for _ in range(50):  # one pose: Stiefel block; one unit sphere: oblique column
        M = rng.uniform(-1, 1, (r, d + 1))
        Y = da.manifold_project(r, d, 1, M)[:, :d]
        assert np.linalg.norm(Y.T @ Y - I) <= 1e-5
        M = rng.uniform(-1, 1, (r, d + 1 + 1))
        s = da.manifold_project(r, d, 1, M, l=1, b=0)[:, d]
        assert abs(np.linalg.norm(s) - 1.0) <= 1e-5
    M = rng.uniform(-1, 1, (r, (d + 1) * n))
    X = da.manifold_project(r, d, n, M)
    assert X.shape == (r, (d + 1) * n)
    for i in range(n):
        Y = X[:, i * (d + 1):i * (d + 1) + d]
        assert np.linalg.norm(Y.T @ Y - I) <= 1e-5
        assert np.array_equal(X[:, i * (d + 1) + d], M[:, i * (d + 1) + d])  # translations are left alone
    M = rng.uniform(-1, 1, (r, (d + 1) * n + l + b))
    X = da.manifold_project(r, d, n, M, l=l, b=b)
    assert X.shape == M.shape
    for i in range(n):
        Y = X[:, i * d:(i + 1) * d]
        assert np.linalg.norm(Y.T @ Y - I) <= 1e-5
    for i in range(l):
        assert abs(np.linalg.norm(X[:, d * n + i]) - 1.0) <= 1e-5
    assert np.array_equal(X[:, d * n + l:], M[:, d * n + l:])
    assert common.rel(X, orc.project_to_manifold(r, d, n, M, l=l, b=b)) < 1e-13


@pytest.mark.parametrize("accel", [True, False])
def test_per_agent_iterate_reproduces_the_session_loop(env, accel):
    """dcora_rbcd_agent_iterate / _agent_get_X / _agent_set_X: the reference's per-agent calls (non-selected agents
    iterate(false), the selected one iterate(true), ref examples/MultiRobotExample.cpp:223-262) give the iterates of
    dcora_rbcd_iterate, across a restart; agents out of step are refused"""
    da, orc = env
    ds = common.product_dataset("smallGrid3D")
    R, r, iters = 5, 5, 34
    X0 = common.random_point(r, ds.d, ds.n, 6, orc.project_to_manifold)
    a = da.RbcdSession(ds, num_robots=R, r=r, acceleration=accel)
    b = da.RbcdSession(ds, num_robots=R, r=r, acceleration=accel)
    per, dh = ds.n // R, ds.d + 1
    for q in range(R):  # Agent::setX block by block
        lo, hi = q * per * dh, (ds.n if q == R - 1 else (q + 1) * per) * dh
        a.agent_set_X(q, X0[:, lo:hi])
        assert a.agent_info(q) == dict(num_poses=(hi - lo) // dh, first_pose=q * per, iteration_number=0)
    b.set_X(X0)
    sel = 0
    for it in range(iters):
        order = [q for q in range(R) if q != sel] + [sel]
        if it % 2:  # the order in which the non-selected agents are called does not matter
            order = order[:-1][::-1] + [sel]
        for q in order:
            a.agent_iterate(q, q == sel)
        c2b, gnb, bnb, nxt = b.iterate(sel)
        c2a, gna, bna, _ = a.evaluate()
        assert abs(c2a - c2b) <= 1e-12 * abs(c2b)
        assert all(a.agent_info(q)["iteration_number"] == it + 1 for q in range(R))
        sel = nxt
    assert np.abs(a.get_X() - b.get_X()).max() < 1e-11
    assert np.abs(np.hstack([a.agent_get_X(q) for q in range(R)]) - a.get_X()).max() == 0
    a.agent_iterate(0, True)
    with pytest.raises(da.DcoraError, match="lockstep"):
        a.agent_iterate(0, True)
