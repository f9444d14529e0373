"""The partitioned sparse inverse on graphs the datasets do not hold (dcora_amd/csrc/host_sparse.cpp: the ordering --
components, threshold cuts, vertex covers of the cut edges, hubs; host_partinv3.cpp: level groups, chained wave
records).  The operator is the reference's preconditioner solve (ref src/Graph.cpp:1901-1917, applied in
src/QuadraticProblem.cpp:70-84); the graphs are the shapes the reference's inputs can take: chains, stars,
several components, isolated variables, lattices with loop closures, variables seen by hundreds of others.

CPU: the host builder's schedule replayed on the host solves the system.
GPU: the device replay equals the oracle's sparse Cholesky solve on the same matrix."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import common
from test_sparse_precond import _selftest, sparse_env  # noqa: F401  (fixture)


def block_laplacian(n, edges, block, rng, reg):
    """sum over edges of [R; -I] W [R; -I]^T with W positive definite, R orthogonal, plus reg I: the shape of
    AbT Omega AbT^T (ref src/Graph.cpp:656-672) with (d+1) x (d+1) blocks"""
    rows, cols, vals = [], [], []
    D = np.zeros((n, block, block))
    ar = np.arange(block)
    for (i, j) in edges:
        B = rng.standard_normal((block, block))
        W = B @ B.T + 0.1 * np.eye(block)
        R = np.linalg.qr(rng.standard_normal((block, block)))[0]
        D[i] += R @ W @ R.T
        D[j] += W
        O = -R @ W
        ri, cj = np.meshgrid(i * block + ar, j * block + ar, indexing="ij")
        rows += [ri.ravel(), cj.ravel()]
        cols += [cj.ravel(), ri.ravel()]
        vals += [O.ravel(), O.ravel()]
    for i in range(n):
        ri, ci = np.meshgrid(i * block + ar, i * block + ar, indexing="ij")
        rows.append(ri.ravel())
        cols.append(ci.ravel())
        vals.append((D[i] + reg * np.eye(block)).ravel())
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n * block, n * block))
    A.sum_duplicates()
    return A


def lattice3(g, rng, keep=0.6, extra=0, hubs=0, hub_degree=200):
    idx = lambda x, y, z: (x * g + y) * g + z
    e = []
    for x in range(g):
        for y in range(g):
            for z in range(g):
                if x + 1 < g:
                    e.append((idx(x, y, z), idx(x + 1, y, z)))
                if y + 1 < g:
                    e.append((idx(x, y, z), idx(x, y + 1, z)))
                if z + 1 < g:
                    e.append((idx(x, y, z), idx(x, y, z + 1)))
    n = g ** 3
    e = [p for p in e if rng.random() < keep and p[1] != p[0] + 1] + [(i, i + 1) for i in range(n - 1)]
    e += [(int(a), int(b)) for a, b in rng.integers(0, n, (extra, 2)) if a != b]
    for h in range(hubs):
        e += [(n + h, int(i)) for i in rng.choice(n, min(n, hub_degree), replace=False)]
    return n + hubs, e


def small_graphs(rng):
    yield "chain", 300, [(i, i + 1) for i in range(299)]
    yield "star", 200, [(0, i) for i in range(1, 200)]
    n = 400
    yield "random", n, [(i, i + 1) for i in range(n - 1)] + [(int(a), int(b)) for a, b in rng.integers(0, n, (300, 2)) if a != b]
    # a chain, a clique, a ring and 20 isolated variables
    e = [(i, i + 1) for i in range(99)] + [(100 + i, 100 + j) for i in range(12) for j in range(i + 1, 12)]
    e += [(120 + i, 120 + (i + 1) % 60) for i in range(60)]
    yield "components", 200, e
    g = 18
    e = [(x * g + y, (x + 1) * g + y) for x in range(g - 1) for y in range(g)]
    e += [(x * g + y, x * g + y + 1) for x in range(g) for y in range(g - 1)]
    yield "grid", g * g, e
    yield "grid with two hubs", g * g + 2, e + [(g * g, i) for i in range(0, g * g, 3)] + [(g * g + 1, i) for i in range(1, g * g, 5)]
    yield "one variable", 1, []
    yield "one edge", 2, [(0, 1)]


@pytest.mark.parametrize("block,r", [(1, 1), (3, 5), (4, 16)])
def test_host_replay_on_irregular_graphs(built, block, r):
    rng = np.random.default_rng(5)
    for name, n, e in small_graphs(rng):
        rc, err, info = _selftest(block_laplacian(n, e, block, rng, 0.1), block, r)
        assert rc == 0 and err < 1e-11, (name, rc, err)


@pytest.mark.parametrize("g,extra,hubs,block,r,min_launches", [(18, 0, 0, 4, 5, 7), (16, 40, 2, 3, 7, 5)])
def test_host_replay_with_level_groups_and_chained_records(built, g, extra, hubs, block, r, min_launches):
    """a 3-D lattice is deep enough for the schedule to merge runs of levels into groups (one launch each way) and for
    tiles of many short runs to continue in chained records; info[0] counts the launches"""
    rng = np.random.default_rng(11)
    n, e = lattice3(g, rng, extra=extra, hubs=hubs)
    rc, err, info = _selftest(block_laplacian(n, e, block, rng, 0.1), block, r)
    assert rc == 0 and err < 1e-11, (rc, err)
    assert info[0] >= min_launches, info


@pytest.mark.gpu
@pytest.mark.parametrize("g,extra,hubs,r", [(14, 0, 0, 5), (18, 0, 0, 5), (16, 40, 2, 7), (16, 0, 1, 3), (12, 500, 0, 12),
                                            # small graphs with hubs: the replay is ONE launch, so the second stage of the
                                            # hubs' correction (x2 from the slices) runs as a launch of its own
                                            (4, 0, 1, 3), (5, 0, 2, 5), (2, 0, 1, 3), (3, 0, 1, 4)])
def test_device_replay_on_irregular_graphs_matches_oracle(sparse_env, g, extra, hubs, r):
    import dcora_amd as da
    from oracle import orc
    rng = np.random.default_rng(100 + g + r)
    d = 3
    n, e = lattice3(g, rng, extra=extra, hubs=hubs)
    Q = block_laplacian(n, e, d + 1, rng, 0.0)
    P = da.QuadraticProblem(r, d, n, da.Csr.from_scipy(Q), reg=0.1)
    Po = orc.Problem(r, d, n, orc.CSR.from_scipy(Q))
    assert P.precond_info()["kind"] == "sparse"
    X = common.random_point(r, d, n, 11, orc.project_to_manifold)
    V = common.random_tangent(r, d, n, 12)
    assert common.rel(P.PreCondition(X, V), Po.precondition(X, V)) < 1e-10
    P.close()
