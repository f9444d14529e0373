"""scratch: certificate at the GPU's rank-4 point of tiers -- device Lanczos vs oracle Lanczos vs scipy"""
import os, sys, time, gzip, shutil, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, cora_flow, dcora_amd as da
from oracle import orc
import scipy.sparse.linalg as sla
ra = da.RADataset(os.path.join(common.DATA, "tiers.pyfg.gz"))
hip = cora_flow.ProductBackend(ra)
X = ra.X_odom
r = ra.d
for level in range(3):
    P = hip.problem(r)
    X, f, gn, outer, inner = hip.optimize(P, X)
    S = da.dual_certificate(r, ra.d, ra.n, X, ra.Q, l=ra.l, b=ra.b)
    t = time.time(); psd, theta, v, lmin = da.fast_verification(S, 1e-4, block=1); t_dev = time.time() - t
    Ss = S.to_scipy()
    So = orc.CSR.from_scipy(Ss)
    t = time.time(); psdo, thetao, vo, lmino = orc.fast_verification(So, 1e-4, block=1); t_cpu = time.time() - t
    t = time.time(); w = sla.eigsh(Ss, k=1, sigma=-1.0, which="LM", return_eigenvectors=False)[0]; t_sp = time.time() - t
    rq = float(v @ (Ss @ v)) if v is not None else None
    print("r", r, "f %.9f gn %.2e" % (f, gn), "| dev psd", psd, "theta", theta, "RQ", rq, "%.2fs" % t_dev,
          "| cpu psd", psdo, "theta", thetao, "%.2fs" % t_cpu, "| scipy lmin(shift-invert)", w, "%.2fs" % t_sp, flush=True)
    if psd:
        break
    Pn = hip.problem(r + 1)
    use_theta, use_v = (theta, v) if theta < 0 else (thetao, vo)
    Xn = Pn.escapeSaddle(X, use_theta, use_v, 1e-4, 1e-4, isSecondOrder=True)
    if Xn is None:
        print("escape failed"); break
    X = Xn; r += 1
