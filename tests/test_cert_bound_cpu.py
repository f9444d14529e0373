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
