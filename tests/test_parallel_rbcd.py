"""Several agents updating at the same time (the asynchronous mode's concurrent Agent::iterate(true), ref
src/Agent.cpp:650-678, as synchronous ticks): dcora_rbcd_iterate_set / dcora_rbcd_agent_colours.

The parity statements need no new oracle: a tick over mutually non-adjacent agents equals the same agents updated
one after the other (dcora_rbcd_iterate, itself checked against the oracle in test_gpu_parity.py), a tick over
adjacent agents equals each agent's solve from the common snapshot, and both are checked against the oracle's
local solver driven from numpy."""
import numpy as np
import pytest
import scipy.sparse as sp

import common


def _agent_graph(ds, R):
    per = ds.n // R
    rb = np.minimum(ds.ids[:, [1, 3]] // per, R - 1)
    adj = [set() for _ in range(R)]
    for a, b in rb[rb[:, 0] != rb[:, 1]]:
        adj[a].add(int(b))
        adj[b].add(int(a))
    return adj


def _sets(col, nc):
    return [np.flatnonzero(col == c).astype(np.int32) for c in range(nc)]


def _oracle_sweep(name, R, r, X, order_sets, sweeps):
    """numpy driver over the oracle's local solver: the agents of a set read one snapshot"""
    from oracle import orc
    ds = common.oracle_dataset(name)
    d, n, dh = ds.d, ds.n, ds.d + 1
    Qg = orc.build_Q_pgo(ds).to_scipy().tocsc()
    per = n // R
    cols = [np.arange(b * per * dh, (n if b == R - 1 else (b + 1) * per) * dh) for b in range(R)]
    X = X.copy()
    for _ in range(sweeps):
        for S in order_sets:
            snap = X.copy()
            for b in S:
                own = cols[b]
                Qcb = Qg[:, own].tolil()
                Qcb[own, :] = 0
                G = snap @ Qcb.tocsc()
                P = orc.Problem(r, d, own.size // dh, orc.CSR.from_scipy(sp.csr_matrix(Qg[own][:, own])), G=G)
                X[:, own] = P.optimize(snap[:, own])[0]
    return X


@pytest.mark.gpu
@pytest.mark.parametrize("name,R", [("sphere2500", 5), ("torus3D", 8), ("smallGrid3D", 5)])
def test_colouring_is_proper(built, name, R):
    import dcora_amd as da
    ds = common.product_dataset(name)
    s = da.RbcdSession(ds, num_robots=R, r=5, acceleration=False)
    col, nc = s.colours()
    adj = _agent_graph(ds, R)
    assert nc == col.max() + 1 and all(col[a] != col[b] for a in range(R) for b in adj[a])
    # greedy in agent order: every agent has the smallest colour its lower-numbered neighbours leave free
    for a in range(R):
        used = {col[b] for b in adj[a] if b < a}
        assert col[a] == min(c for c in range(R) if c not in used)


@pytest.mark.gpu
@pytest.mark.parametrize("name,R,r", [("sphere2500", 5, 5), ("torus3D", 8, 5), ("torus3D", 8, 9)])
def test_coloured_tick_equals_one_after_the_other(built, name, R, r):
    import dcora_amd as da
    from oracle import orc
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 11, orc.project_to_manifold)
    par = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    seq = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    par.set_X(X0)
    seq.set_X(X0)
    col, nc = par.colours()
    sets = _sets(col, nc)
    assert max(len(S) for S in sets) >= 2  # something does run concurrently
    costs = []
    for sweep in range(3):
        for S in sets:
            par.iterate_set(S)
            for a in S:
                c2s = seq.iterate(int(a))[0]
        c2p = par.evaluate()[0]
        costs.append(c2p)
        assert abs(c2p - c2s) <= 1e-12 * abs(c2s)
        assert np.abs(par.get_X() - seq.get_X()).max() < 1e-11
    assert costs[2] < costs[1] < costs[0]  # non-adjacent exact block updates: monotone
    if name == "sphere2500":
        Xo = _oracle_sweep(name, R, r, X0, sets, 1)
        par.set_X(X0)
        for S in sets:
            par.iterate_set(S)
        ok = orc.Problem(r, ds.d, ds.n, orc.build_Q_pgo(common.oracle_dataset(name)), reg=-1)
        fo, fp = ok.f(Xo), ok.f(par.get_X())
        assert abs(fo - fp) < 1e-7 * abs(fo)


@pytest.mark.gpu
def test_simultaneous_adjacent_agents_read_one_snapshot(built):
    import dcora_amd as da
    from oracle import orc
    name, R, r = "smallGrid3D", 5, 5
    ds = common.product_dataset(name)
    X0 = common.random_point(r, ds.d, ds.n, 5, orc.project_to_manifold)
    s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    one = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    everyone = np.arange(R, dtype=np.int32)
    s.set_X(X0)
    with pytest.raises(da.DcoraError, match="share measurements"):
        s.iterate_set(everyone)
    s.iterate_set(everyone, allow_adjacent=True)
    Xs = s.get_X()
    per, dh = ds.n // R, ds.d + 1
    for a in range(R):
        one.set_X(X0)
        one.iterate(a)
        own = slice(a * per * dh, (ds.n if a == R - 1 else (a + 1) * per) * dh)
        assert np.abs(one.get_X()[:, own] - Xs[:, own]).max() < 1e-11
    Xo = _oracle_sweep(name, R, r, X0, [everyone], 1)
    assert np.abs(Xo - Xs).max() < 1e-6


@pytest.mark.gpu
def test_simultaneous_updates_refuse_acceleration_and_bad_sets(built):
    import dcora_amd as da
    ds = common.product_dataset("smallGrid3D")
    acc = da.RbcdSession(ds, num_robots=5, r=5, acceleration=True)
    with pytest.raises(da.DcoraError, match="acceleration off"):
        acc.iterate_set([0, 2])
    s = da.RbcdSession(ds, num_robots=5, r=5, acceleration=False)
    for bad in ([0, 0], [5], [-1]):
        with pytest.raises(da.DcoraError, match="out of range or twice"):
            s.iterate_set(bad)


def test_oracle_coloured_ticks_do_not_depend_on_the_thread_count(built):
    """the CPU side of the coloured mode (oracle run_coloured: one host thread per agent of a colour, ref
    src/Agent.cpp:660-662) is the same arithmetic on 1 and on R threads, and its cost decreases monotonically"""
    from oracle import orc
    dso = common.oracle_dataset("smallGrid3D")
    X0 = common.random_point(5, dso.d, dso.n, 7, orc.project_to_manifold)
    a = orc.run_coloured(dso, X0, num_robots=5, r=5, sweeps=6, threads=1)
    b = orc.run_coloured(dso, X0, num_robots=5, r=5, sweeps=6, threads=5)
    assert a["colours"] == b["colours"] == 2
    assert np.array_equal(a["cost"], b["cost"]) and np.array_equal(a["X"], b["X"])
    assert np.all(np.diff(a["cost"]) <= 1e-9 * a["cost"][:-1])


@pytest.mark.gpu
@pytest.mark.parametrize("name,R", [("sphere2500", 5), ("torus3D", 8)])
def test_coloured_sweeps_match_the_oracles_threaded_agents(built, name, R):
    """dcora_rbcd_iterate_set over the colours against the oracle's own Agents updating colour by colour on R host
    threads: same cost after every sweep"""
    import dcora_amd as da
    from oracle import orc
    ds, dso = common.product_dataset(name), common.oracle_dataset(name)
    r, sweeps = 5, 4
    X0 = common.random_point(r, ds.d, ds.n, 5, orc.project_to_manifold)
    want = orc.run_coloured(dso, X0, num_robots=R, r=r, sweeps=sweeps, threads=R)
    s = da.RbcdSession(ds, num_robots=R, r=r, acceleration=False)
    col, nc = s.colours()
    assert nc == want["colours"]
    s.set_X(X0)
    cost = []
    for _ in range(sweeps):
        for S in _sets(col, nc):
            s.iterate_set(S)
        cost.append(s.evaluate()[0])
    X = s.get_X()
    s.close()
    assert np.allclose(cost, want["cost"], rtol=1e-8, atol=0), (cost, want["cost"])
    assert common.rel(X, want["X"]) < 1e-6
