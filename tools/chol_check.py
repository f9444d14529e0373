"""device multifrontal Cholesky (dcora_cert_is_psd_device) against scipy on the product's matrices: verdict, log det,
timing.  python tools/chol_check.py [--big]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da  # noqa: E402
from dcora_amd import datasets  # noqa: E402


def logdet_ref(A):
    lu = spl.splu(sp.csc_matrix(A), permc_spec="COLAMD", diag_pivot_thresh=0.0)
    d = lu.U.diagonal()
    return float(np.sum(np.log(np.abs(d)))), bool(np.all(d > 0))


def run(A, block, name, ref=True, host=True):
    S = da.Csr.from_scipy(sp.csr_matrix(A))
    da.chol_cache_clear()
    t = time.perf_counter()
    ok, info = da.is_psd_device(S, block, info=True)
    cold = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter()
    ok2, info2 = da.is_psd_device(S, block, info=True)
    warm = 1e3 * (time.perf_counter() - t)
    t = time.perf_counter()
    okh = da.is_psd(S, block) if host else None
    host = 1e3 * (time.perf_counter() - t)
    line = "%-28s n %7d pd %s/%s host %s | cold %.1f ms (symbolic %.1f) warm %.2f ms (numeric %.2f, look-up %.2f) host %.1f ms | " \
           "levels %d launches %d arena %.1f MB %.2f Gflop" % (
               name, S.n, ok, ok2, okh, cold, info["symbolic_ms"], warm, info2["numeric_ms"], info2["lookup_ms"], host,
               info["levels"], info["launches"], info["arena_bytes"] / 1e6, info["flops"] / 1e9)
    if ref and ok:
        ld, pos = logdet_ref(A)
        line += " | logdet %.10g ref %.10g pos %s" % (info["logdet"], ld, pos)
    print(line, flush=True)


def main():
    big = "--big" in sys.argv
    for name in ["smallGrid3D", "sphere2500"]:
        ds = datasets.product_dataset(name)
        Q = da.build_Q_pgo(ds).to_scipy()
        n = Q.shape[0]
        run(Q + sp.identity(n), ds.d + 1, name + " Q+I")
        run(Q - 0.5 * sp.identity(n), ds.d + 1, name + " Q-0.5I", ref=False)
        run(Q + 1e-3 * sp.identity(n), ds.d + 1, name + " Q+1e-3I")
    ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
    Q = ra.Q.to_scipy()
    run(Q + sp.identity(Q.shape[0]), 1, "tiers Q+I")
    run(Q - 1e-2 * sp.identity(Q.shape[0]), 1, "tiers Q-0.01I", ref=False)
    if big:
        from dcora_amd import synth
        for dims in ((20, 20, 20), (30, 30, 30), (50, 50, 40)):
            ds = synth.lattice_se3(*dims)
            Q = da.build_Q_pgo(ds).to_scipy()
            run(Q + sp.identity(Q.shape[0]), 4, "lattice %dx%dx%d Q+I" % dims, ref=False, host=dims[0] < 50)


if __name__ == "__main__":
    main()
