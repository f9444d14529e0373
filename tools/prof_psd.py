"""scratch: the PSD test (numeric factorisation on the device) of the whole 100k lattice / sphere2500, three repeats"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scipy.sparse as sp
import dcora_amd as da
from dcora_amd import synth
import common
which = sys.argv[1] if len(sys.argv) > 1 else "lattice"
ds, blk = (synth.lattice_se3(), 4) if which == "lattice" else (common.product_dataset("sphere2500"), 4)
Q = da.build_Q_pgo(ds).to_scipy()
A = da.Csr.from_scipy((Q + 1e-3 * sp.identity(Q.shape[0])).tocsr())
for rep in range(4):
    t0 = time.perf_counter()
    ok, info = da.is_psd_device(A, blk, info=True)
    ms = 1e3 * (time.perf_counter() - t0)
    print("%s rep %d: %s, %.2f ms (numeric %.2f ms, %.1f Gflop -> %.2f Tflop/s, %d launches)" % (
        which, rep, ok, ms, info["numeric_ms"], info["flops"] / 1e9, info["flops"] / info["numeric_ms"] / 1e9, info["launches"]), flush=True)
