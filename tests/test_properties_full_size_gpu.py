"""Size-independent properties of the operators at BASELINE.json's FULL sizes, where the CPU oracle is too slow to be
the checker: the whole 100k-pose lattice (k = 400 000), one of its agent blocks (k = 50 000), sphere2500 as one problem
and tiers.pyfg (range-aided layout, one hub).  The operators are the reference's (src/QuadraticProblem.cpp:38-84):

  * the preconditioner inverts what the Q-apply applies:  Proj_X((Q + reg I)^-1 (W (Q + reg I))) = Proj_X(W) for ANY W
    -- the sparse replay (27 launches on the whole lattice) against the block Q-apply, no third party involved;
  * Hessian and preconditioner are linear and self-adjoint on the tangent space, the preconditioner positive definite;
  * the tangent projection is idempotent and orthogonal to Y A for symmetric A;  Retract(Y, 0) = Y;  Y^T Y = I after a
    retraction and after the metric projection;
  * the translation gauge is in the null space of Q (SURVEY 8(c) iii).
"""
import os

import numpy as np
import pytest

import common
from test_raslam import ra_path

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(built):
    import dcora_amd as da
    if da.device_count() < 1:
        pytest.fail("no GPU visible: the product has no CPU fallback")
    return da


def _agent_block(ds, R, b):
    """measurement arrays of agent b under the reference driver's contiguous partition
    (ref examples/MultiRobotExample.cpp:56-83): local pose indices, robot ids"""
    per = ds.n // R
    robot = np.minimum(ds.ids[:, [1, 3]] // per, R - 1)
    start = robot * per
    ids = ds.ids.copy()
    ids[:, 0], ids[:, 2] = robot[:, 0], robot[:, 1]
    ids[:, 1] -= start[:, 0]
    ids[:, 3] -= start[:, 1]
    keep = (robot[:, 0] == b) | (robot[:, 1] == b)
    nb = ds.n - b * per if b == R - 1 else per
    return nb, ids[keep], ds.vals[keep]


def _cases(da, which):
    """-> (problem, r, d, n, l, b, k, reg, manifold_project)"""
    from dcora_amd import synth
    if which == "lattice100k":
        ds = synth.lattice_se3()
        return da.QuadraticProblem(5, ds.d, ds.n, da.build_Q_pgo(ds), reg=0.1), 5, ds.d, ds.n, 0, 0, 0.1
    if which == "lattice_agent":
        ds = synth.lattice_se3()
        nb, ids, vals = _agent_block(ds, 8, 3)
        Q = da.build_Q_pgo(ds, n=nb, agent=3, ids=ids, vals=vals)
        return da.QuadraticProblem(7, ds.d, nb, Q, reg=0.1), 7, ds.d, nb, 0, 0, 0.1
    if which == "sphere2500":
        ds = common.product_dataset("sphere2500")
        return da.QuadraticProblem(5, ds.d, ds.n, da.build_Q_pgo(ds), reg=0.1), 5, ds.d, ds.n, 0, 0, 0.1
    ra = da.RADataset(ra_path("tiers"))
    reg = da.precond_regularization(ra.Q)
    return da.QuadraticProblem(3, ra.d, ra.n, ra.Q, reg=reg, l=ra.l, b=ra.b), 3, ra.d, ra.n, ra.l, ra.b, reg


def _dot(a, b):
    return float(np.vdot(a, b))


@pytest.mark.parametrize("which", ["sphere2500", "lattice_agent", "lattice100k", "tiers"])
def test_operator_identities_at_full_size(env, which):
    da = env
    P, r, d, n, l, b, reg = _cases(da, which)
    k = (d + 1) * n + l + b
    rng = np.random.default_rng(31)
    X = da.manifold_project(r, d, n, rng.standard_normal((r, k)), l=l, b=b)
    tang = lambda V: P.projectToTangentSpace(X, V)
    U, V, W = (rng.standard_normal((r, k)) for _ in range(3))
    # the preconditioner undoes the Q-apply (EucGrad without a linear term is W -> W Q, for any W)
    AW = P.EucGrad(W) + reg * W
    assert common.rel(P.PreCondition(X, AW), tang(W)) < 1e-8
    Ut, Vt = tang(U), tang(V)
    # idempotence and orthogonality of the tangent projection (rotation blocks: <Proj V, Y A> = 0 for symmetric A)
    assert common.rel(tang(Ut), Ut) < 1e-13
    if l == 0 and b == 0:
        A = rng.standard_normal((d, d))
        A = A + A.T
        Xb = X.reshape(r, n, d + 1)
        YA = np.zeros_like(Xb)
        YA[:, :, :d] = np.einsum("rnc,ca->rna", Xb[:, :, :d], A)
        assert abs(_dot(Ut, YA.reshape(r, k))) < 1e-9 * np.linalg.norm(Ut) * np.linalg.norm(YA)
    # linear, self-adjoint, positive definite
    HU, HV = P.HessVec(X, Ut), P.HessVec(X, Vt)
    assert common.rel(P.HessVec(X, 2.0 * Ut - 3.0 * Vt), 2.0 * HU - 3.0 * HV) < 1e-12
    assert abs(_dot(Ut, HV) - _dot(Vt, HU)) < 1e-10 * (np.linalg.norm(Ut) * np.linalg.norm(HV))
    PU, PV = P.PreCondition(X, Ut), P.PreCondition(X, Vt)
    assert common.rel(P.PreCondition(X, 2.0 * Ut - 3.0 * Vt), 2.0 * PU - 3.0 * PV) < 1e-10
    assert abs(_dot(Ut, PV) - _dot(Vt, PU)) < 1e-9 * (np.linalg.norm(Ut) * np.linalg.norm(PV))
    assert _dot(Ut, PU) > 0 and _dot(Vt, PV) > 0
    # retraction and metric projection land on the manifold; a zero step stays
    assert common.rel(P.Retract(X, np.zeros_like(X)), X) < 1e-14
    Z = P.Retract(X, 0.5 * Ut)
    M = da.manifold_project(r, d, n, X + 0.3 * U, l=l, b=b)
    for Y in (Z, M):
        if l == 0 and b == 0:
            B = Y.reshape(r, n, d + 1)[:, :, :d]
            G = np.einsum("rna,rnc->nac", B, B)
        else:
            B = Y[:, :d * n].reshape(r, n, d)
            G = np.einsum("rna,rnc->nac", B, B)
            S = Y[:, d * n:d * n + l]
            assert np.abs(np.sum(S * S, axis=0) - 1.0).max() < 1e-13
        assert np.abs(G - np.eye(d)).max() < 1e-12
    P.close()


@pytest.mark.parametrize("which", ["sphere2500", "lattice100k"])
def test_translation_gauge_is_in_the_null_space_of_Q(env, which):
    da = env
    P, r, d, n, l, b, reg = _cases(da, which)
    k = (d + 1) * n
    rng = np.random.default_rng(5)
    T = np.zeros((r, k))
    T[:, d::d + 1] = rng.standard_normal((r, 1))  # the same translation for every pose, no rotation part
    G = P.EucGrad(T)
    assert np.abs(G).max() < 1e-9 * np.abs(T).max() * 1e3
    P.close()
