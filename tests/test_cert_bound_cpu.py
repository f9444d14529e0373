"""dcora_cert_lambda_min_certified (an addition of this library, include/dcora_hip.h) returns a VERIFIED lower bound of
lambda_min(S), not an estimate: host code, runs without a GPU.  S = Q - mu I with Q the connection Laplacian of a pose
graph (positive semidefinite with a null space: ref src/Graph.cpp:579-683), so lambda_min(S) = -mu exactly."""
import numpy as np
import pytest
import scipy.sparse as sp

import common


@pytest.mark.parametrize("name,mu", [("smallGrid3D", 5e-4), ("tinyGrid3D", 1e-6), ("sphere2500", 8.4e-6)])
def test_bound_is_below_and_close_to_the_smallest_eigenvalue(built, name, mu):
    import dcora_amd as da
    ds = common.product_dataset(name)
    Q = da.build_Q_pgo(ds).to_scipy()
    S = da.Csr.from_scipy(Q - mu * sp.identity(Q.shape[0]))
    lam, its = da.lambda_min_certified(S, 1e-3, block=ds.d + 1)
    assert lam <= -mu + 1e-13, (lam, mu)           # a bound ...
    assert -mu - lam < 5e-2 * mu + 1e-9, (lam, mu)  # ... that is tight (1 / theta - eta amplifies by eta / mu)
    assert lam > -1e-3 and its > 0


def test_unaccepted_certificate_is_refused(built):
    import dcora_amd as da
    ds = common.product_dataset("tinyGrid3D")
    Q = da.build_Q_pgo(ds).to_scipy()
    S = da.Csr.from_scipy(Q - 1e-2 * sp.identity(Q.shape[0]))
    with pytest.raises(Exception):
        da.lambda_min_certified(S, 1e-3, block=ds.d + 1)


def test_oracle_min_eig_on_matrices_whose_krylov_space_is_exhausted(built):
    """the oracle's Lanczos (the checker of dcora_cert_min_eig) on a zero matrix, on 2 I and on a matrix with three
    distinct eigenvalues: the Krylov space ends after 0, 1 and 3 steps and the run must notice -- relative to |S v_j|,
    the remainder of the orthogonalisation is rounding noise there, not a direction (with an absolute 1e-300 alone the
    noise was normalised into the basis: 2 I came out as -1.02).  The product's two forms of the cycle are checked on
    the same matrices in tests/test_kernel_forms_gpu.py."""
    from oracle import orc
    n = 400
    zero = sp.csr_matrix((np.zeros(n), (np.arange(n), np.arange(n))), shape=(n, n))
    ok, lam, v, mv = orc.min_eig(orc.CSR.from_scipy(zero), tol=1e-6)
    assert ok and lam == 0.0 and abs(np.linalg.norm(v) - 1) < 1e-9
    ok, lam, v, mv = orc.min_eig(orc.CSR.from_scipy((2.0 * sp.identity(n)).tocsr()), tol=1e-6)
    assert ok and abs(lam - 2.0) < 1e-12 and mv < 200
    D = sp.diags(np.r_[-2.0, np.ones(200), 3.0 * np.ones(199)]).tocsr()
    ok, lam, v, mv = orc.min_eig(orc.CSR.from_scipy(D), tol=1e-8)
    assert ok and abs(lam + 2.0) < 1e-7 and abs(abs(v[0]) - 1.0) < 1e-6
