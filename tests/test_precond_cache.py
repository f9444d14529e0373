"""The preconditioner image is cached inside the library, keyed on the content of Q + reg I (SURVEY.md section 8b,
Ownership): the reference re-creates its QuadraticProblem on every Agent::updateX (ref src/Agent.cpp:1252) while its
Graph keeps Q and the factor (ref src/Graph.cpp:523-533, 1901-1917).  A problem created again on the same matrix --
at another rank, after the first one was destroyed -- must attach to the resident image, give bitwise the same
operator, and cost milliseconds instead of a factorisation."""
import os
import time

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


def _agent_Q(da, ds, R, b=0):
    import bench
    nb, ids, vals = bench.agent_block(ds, R, b)
    return nb, da.build_Q_pgo(ds, n=nb, agent=b, ids=ids, vals=vals)


@pytest.mark.parametrize("R,kind", [(5, "dense"), (1, "sparse")])
def test_second_problem_on_the_same_Q_attaches_to_the_cached_inverse(env, R, kind):
    da, orc = env
    if os.environ.get("DCORA_PRECOND", kind) != kind:
        pytest.skip("DCORA_PRECOND overrides the choice of preconditioner this case is about")
    ds = common.product_dataset("sphere2500")
    nb, Q = _agent_Q(da, ds, R)
    k = 4 * nb
    da.precond_cache_clear()
    c0 = da.precond_cache_info()
    t0 = time.perf_counter()
    P5 = da.QuadraticProblem(5, 3, nb, Q)
    t_first = time.perf_counter() - t0
    assert P5.precond_info()["kind"] == kind
    X5 = common.random_point(5, 3, nb, 1, orc.project_to_manifold)
    V5 = orc.tangent_project(5, 3, nb, X5, common.random_tangent(5, 3, nb, 2))
    Z5 = P5.PreCondition(X5, V5)
    # the next staircase level: same Q, rank 6
    t0 = time.perf_counter()
    P6 = da.QuadraticProblem(6, 3, nb, Q)
    t_second = time.perf_counter() - t0
    c1 = da.precond_cache_info()
    assert c1["hits"] == c0["hits"] + 1 and c1["misses"] == c0["misses"] + 1 and c1["entries"] == 1
    assert P6.precond_info()["setup_ms"] < 0.2 * P5.precond_info()["setup_ms"]
    assert t_second < 0.5 * t_first
    # same operator: rank-5 vectors padded with a zero row give the rank-5 result in the first five rows
    X6 = np.vstack([X5, np.zeros((1, k))])
    V6 = np.vstack([V5, np.zeros((1, k))])
    Z6 = P6.PreCondition(X6, V6)
    if kind == "dense":
        assert np.array_equal(Z6[:5], Z5) and not Z6[5].any()
    else:  # the level replay groups its lanes by r: another summation order at another rank
        assert common.rel(Z6[:5], Z5) < 1e-12 and not Z6[5].any()
    # the image outlives the problem that built it
    P5.close()
    assert np.array_equal(P6.PreCondition(X6, V6), Z6)
    # and a problem built WITHOUT the cache gives bitwise the same operator
    P6.close()
    da.precond_cache_clear()
    P5b = da.QuadraticProblem(5, 3, nb, Q)
    assert np.array_equal(P5b.PreCondition(X5, V5), Z5)
    # a different matrix (one value changed) or another regularisation is another entry
    Q2 = da.Csr(Q.n, Q.rp.copy(), Q.ci.copy(), Q.v.copy())
    Q2.v[0] *= 1.0 + 1e-12
    m0 = da.precond_cache_info()["misses"]
    da.QuadraticProblem(5, 3, nb, Q2).close()
    da.QuadraticProblem(5, 3, nb, Q, reg=0.2).close()
    assert da.precond_cache_info()["misses"] == m0 + 2
    P5b.close()
    print("%s inverse, k = %d: first create %.1f ms, second (cached) %.2f ms" % (kind, k, 1e3 * t_first, 1e3 * t_second))


def test_staircase_levels_share_the_agents_inverses(env):
    """the driver re-creates the session at every staircase level (ref examples/MultiRobotExample.cpp:172-217): the
    second session's agents hit the cache"""
    da, orc = env
    ds = common.product_dataset("smallGrid3D")
    da.precond_cache_clear()
    z = da.precond_cache_info()
    s5 = da.RbcdSession(ds, num_robots=5, r=5)
    a = da.precond_cache_info()
    assert a["misses"] == z["misses"] + 5 and a["hits"] == z["hits"]
    s6 = da.RbcdSession(ds, num_robots=5, r=6)
    b = da.precond_cache_info()
    assert b["hits"] == a["hits"] + 5 and b["misses"] == a["misses"] and b["entries"] == 5
    # both sessions run
    X5 = common.random_point(5, 3, ds.n, 3, orc.project_to_manifold)
    s5.set_X(X5)
    s6.set_X(np.vstack([X5, np.zeros((1, 4 * ds.n))]))
    o5, o6 = s5.run(max_iters=5, rgrad_tol=0.0), s6.run(max_iters=5, rgrad_tol=0.0)
    assert o5["cost"][-1] < o5["cost"][0] and o6["cost"][-1] < o6["cost"][0]
    s5.close()
    s6.close()
