"""timings of the pieces of BASELINE config 5 at full size (100k-pose lattice, 8 agents, staircase)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
from dcora_amd import synth
t0 = time.perf_counter(); ds = synth.lattice_se3(); print("lattice %.2fs n=%d m=%d" % (time.perf_counter() - t0, ds.n, ds.m), flush=True)
r, R = 5, 8
t0 = time.perf_counter(); T = da.chordal_initialization(ds); print("chordal %.2fs" % (time.perf_counter() - t0), flush=True)
X0 = np.zeros((r, 4 * ds.n)); X0[:3] = T
t0 = time.perf_counter(); s = da.RbcdSession(ds, num_robots=R, r=r); print("session %.2fs" % (time.perf_counter() - t0), flush=True)
s.set_X(X0)
t0 = time.perf_counter(); out = s.run(max_iters=int(sys.argv[1]) if len(sys.argv) > 1 else 1500, rgrad_tol=0.1)
dt = time.perf_counter() - t0
print("rbcd %d its %.2fs (%.0f it/s) cost %.4f gn %.4f" % (out["iters"], dt, out["iters"] / dt, out["cost"][-1], out["gradnorm"][-1]), flush=True)
print("gn trace", out["gradnorm"][::100], flush=True)
X = s.get_X(); s.close()
t0 = time.perf_counter(); Q = da.build_Q_pgo(ds); print("Q %.2fs nnz %d" % (time.perf_counter() - t0, Q.nnz), flush=True)
t0 = time.perf_counter(); S = da.dual_certificate(r, ds.d, ds.n, X, Q); print("S %.2fs" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); psd = da.is_psd(S, block=4); print("is_psd(S) %s %.2fs" % (psd, time.perf_counter() - t0), flush=True)
t0 = time.perf_counter(); psd, theta, v, lmin = da.fast_verification(S, 1e-3, block=4); print("fastVerification psd=%s theta=%.3e lmin=%.3e %.2fs" % (psd, theta, lmin, time.perf_counter() - t0), flush=True)
if not psd:
    t0 = time.perf_counter(); P6 = da.QuadraticProblem(r + 1, ds.d, ds.n, Q); print("central problem r=6 %.2fs %s" % (time.perf_counter() - t0, P6.precond_info()), flush=True)
    t0 = time.perf_counter(); Xn = P6.escapeSaddle(X, theta, v); print("escape %s %.2fs" % (Xn is not None, time.perf_counter() - t0), flush=True)
