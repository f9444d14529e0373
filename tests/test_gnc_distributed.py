"""The agents' robust outer loop (GNC-TLS weight updates between RBCD rounds, ref src/Agent.cpp:1280-1441) over the
RBCD session: smallGrid3D with gross outlier loop closures added.  The same control flow runs on the CPU oracle
(its RBCD driver, residuals and RobustCost); both must reject exactly the injected outliers and end at the same point."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _with_outliers(ds_cls, base, n_out, seed):
    rng = np.random.default_rng(seed)
    d, n = base.d, base.n
    ids, vals = [base.ids], [base.vals]
    for _ in range(n_out):
        i = int(rng.integers(0, n - 10))
        j = int(rng.integers(i + 5, n))
        Q = np.linalg.qr(rng.standard_normal((d, d)))[0]
        if np.linalg.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        row = np.concatenate([Q.reshape(-1, order="F"), 5.0 * rng.standard_normal(d), [12.5, 100.0, 1.0]])
        ids.append(np.array([[0, i, 0, j]], np.int32))
        vals.append(row[None, :])
    return ds_cls(d, n, np.vstack(ids), np.vstack(vals))


GNC = dict(GNCBarc=10.0, GNCMuStep=2.0)  # residuals of the clean closures stay below 10; 20 updates reach mu ~ 100


def _oracle_flow(orc, ds, X0, R, r, lc, num_weight_updates, inner_iters, rgrad_tol):
    w = ds.vals[:, -1]
    w[lc] = 1.0
    X = X0
    for u in range(num_weight_updates):
        tr = orc.run_rbcd(ds, X, num_robots=R, r_min=r, max_iters=inner_iters, staircase=0, rgrad_tol=rgrad_tol)
        X = tr["X"]
        e = orc.measurement_errors(ds, X)
        w[lc] = orc.robust_weights(np.sqrt(e[lc]), updates=u, cost_type="GNC_TLS", **GNC)
    tr = orc.run_rbcd(ds, X, num_robots=R, r_min=r, max_iters=1000, staircase=0, rgrad_tol=rgrad_tol)
    return tr["X"], w.copy(), tr["cost"][-1]


def test_distributed_gnc_rejects_injected_outliers(built):
    import dcora_amd as da
    from dcora_amd import driver
    from oracle import orc
    if da.device_count() < 1:
        pytest.fail("no GPU visible: the product has no CPU fallback")
    R, r, n_out = 5, 5, 12
    clean = common.product_dataset("smallGrid3D")
    ds = _with_outliers(da.Dataset, clean, n_out, seed=2)
    dso = _with_outliers(orc.Dataset, common.oracle_dataset("smallGrid3D"), n_out, seed=2)
    assert np.array_equal(ds.ids, dso.ids) and np.abs(ds.vals - dso.vals).max() < 1e-13  # two readers, same file
    T = da.chordal_initialization(clean)  # an outlier-free start, as odometry-based initialisation gives
    X0 = np.zeros((r, 4 * ds.n))
    X0[:3] = T
    from dcora_amd import robust as rb
    out = driver.multi_robot_gnc_example(ds, X0, num_robots=R, r=r, robust=rb.RobustCostParameters("GNC_TLS", **GNC),
                                         num_weight_updates=20, inner_iters=30, rgrad_tol=0.1)
    lc, w = out["loop_closures"], out["weights"]
    m0 = clean.m
    assert np.all(w[m0:] < 1e-8), "every injected closure is rejected"
    assert np.all(w[:m0][lc[:m0]] > 1 - 1e-8), "every original closure is kept"
    assert np.all(w[~lc] == 1.0)
    # the robust solution is the optimum of the clean problem: 2 f = 1025.398 (SE-Sync's optimum of smallGrid3D)
    assert abs(out["final"]["cost_2f"] - 1025.398) < 0.05
    assert [rd["rejected"] for rd in out["rounds"]][-1] == n_out
    Xo, wo, co = _oracle_flow(orc, dso, X0, R, r, lc, 20, 30, 0.1)
    assert np.array_equal(w > 0.5, wo > 0.5) and np.abs(w - wo).max() < 1e-6
    assert abs(out["final"]["cost_2f"] - co) < 1e-6 * abs(co)
    # the iterates themselves are only compared through the cost: RBCD stops at |rgrad| < 0.1, where the point is
    # not unique to the digits the two paths agree on
    Qo = orc.build_Q_pgo(dso)  # with the final weights
    assert abs(2 * orc.Problem(r, ds.d, ds.n, Qo).f(out["X"]) - co) < 1e-6 * abs(co)
