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
