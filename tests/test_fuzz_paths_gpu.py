"""Differential test of the solver paths on random problems (a short form of tools/fuzz_paths.py): the same lattice solved
by the default kernels, by the generic path (DCORA_SOLVER=generic, read when a problem is created) and with the sparse
preconditioner forced must give the same iteration counts and iterates, and the solver's own cost bookkeeping must be
the cost of its iterates (scipy)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SWITCHES = ("DCORA_SOLVER", "DCORA_PRECOND")


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_paths_agree_on_a_random_lattice(built, seed):
    import dcora_amd as da
    from dcora_amd import synth
    rng = np.random.default_rng(seed)
    dims = tuple(int(x) for x in rng.integers(2, 13, 3))
    r = int(rng.integers(3, 9))
    ds = synth.lattice_se3(*dims, seed=int(rng.integers(1, 1 << 30)))
    n, k = ds.n, 4 * ds.n
    Q = da.build_Q_pgo(ds)
    A = Q.to_scipy()
    X0 = da.manifold_project(r, 3, n, rng.uniform(-1, 1, (r, k)))
    G = rng.standard_normal((r, k)) * float(rng.choice([0.0, 1.0, 30.0]))
    withG = bool(np.any(G))
    f = lambda Y: 0.5 * float(np.sum((A @ Y.T).T * Y)) + float(np.sum(Y * G))
    saved = {s: os.environ.get(s) for s in SWITCHES}
    outs = {}
    try:
        for tag, env in (("default", {}), ("generic", {"DCORA_SOLVER": "generic"}), ("sparse", {"DCORA_PRECOND": "sparse"}),
                         ("generic+sparse", {"DCORA_SOLVER": "generic", "DCORA_PRECOND": "sparse"})):
            for s in SWITCHES:
                os.environ.pop(s, None)
            os.environ.update(env)
            P = da.QuadraticProblem(r, 3, n, Q, G=G if withG else None)
            opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=3, RTR_tCG_iterations=30, gradnorm_tol=1e-2))
            X = opt.optimize(X0)
            res = opt.getOptResult()
            P.close()
            scale = max(1.0, abs(f(X0)))
            assert abs(res["fInit"] - f(X0)) <= 1e-10 * scale, (tag, dims, r)
            assert abs(res["fOpt"] - f(X)) <= 1e-10 * scale, (tag, dims, r)
            outs[tag] = (X, res)
    finally:
        for s, v in saved.items():
            if v is None:
                os.environ.pop(s, None)
            else:
                os.environ[s] = v
    Xr, rr = outs["default"]
    for tag in ("generic", "sparse", "generic+sparse"):
        X, res = outs[tag]
        assert (res["outer_iterations"], res["inner_iterations"]) == (rr["outer_iterations"], rr["inner_iterations"]), tag
        assert np.linalg.norm(X - Xr) <= 1e-9 * np.linalg.norm(Xr), tag
