"""scratch: per-level breakdown of the tiers.pyfg staircase (problem, RTR, certificate, escape)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
from dcora_amd import cora_flow, datasets
ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
hip = cora_flow.ProductBackend(ra)
X = np.array(ra.X_odom); r = ra.d
while r < 8:
    t0 = time.perf_counter(); P = hip.problem(r); t1 = time.perf_counter()
    Xopt, f, gn, outer, inner = hip.optimize(P, X); t2 = time.perf_counter()
    S = da.dual_certificate(r, ra.d, ra.n, Xopt, hip.Q, l=ra.l, b=ra.b); t3 = time.perf_counter()
    psd, theta, v, lmin = da.fast_verification(S, cora_flow.MIN_EIG_TOL, block=1); t4 = time.perf_counter()
    print("r %d: problem %.0f ms, RTR %.0f ms (%d tCG), dual certificate %.0f ms, fast verification %.0f ms (psd %s theta %.3e)" % (
        r, 1e3*(t1-t0), 1e3*(t2-t1), inner, 1e3*(t3-t2), 1e3*(t4-t3), psd, theta), flush=True)
    if psd: break
    Pn = hip.problem(r + 1); t5 = time.perf_counter()
    Xn = hip.escape(Pn, Xopt, theta, v); t6 = time.perf_counter()
    print("      escape: problem %.0f ms, escapeSaddle %.0f ms" % (1e3*(t5-t4), 1e3*(t6-t5)), flush=True)
    hip.close(Pn); hip.close(P); X = Xn; r += 1
