"""scratch: RTR on tiers at rank d, GPU vs oracle, few outer iterations"""
import os, sys, time, gzip, shutil, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common, dcora_amd as da
from oracle import orc
name = sys.argv[1] if len(sys.argv) > 1 else "tiers"
path = os.path.join(common.DATA, name + ".pyfg.gz")
ra = da.RADataset(path)
fd, tmp = tempfile.mkstemp(suffix=".pyfg")
with os.fdopen(fd, "wb") as out, gzip.open(path, "rb") as src:
    shutil.copyfileobj(src, out)
ro = orc.RADataset(tmp)
reg = da.precond_regularization(ra.Q)
d = ra.d
P = da.QuadraticProblem(d, d, ra.n, ra.Q, reg=reg, l=ra.l, b=ra.b)
Po = orc.Problem(d, d, ra.n, ro.Q, reg=reg, l=ra.l, b=ra.b)
X0 = ra.X_odom
V = np.random.default_rng(0).standard_normal(X0.shape)
print("f", P.f(X0), Po.f(X0))
print("precond rel", common.rel(P.PreCondition(X0, V), Po.precondition(X0, V)))
print("hess rel", common.rel(P.Hess(X0, V) if hasattr(P, "Hess") else Po.hess(X0, V), Po.hess(X0, V)))
print("rgrad rel", common.rel(P.RieGrad(X0), Po.rgrad(X0)))
for outer in (1, 2, 3, 6):
    opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=outer, RTR_tCG_iterations=200, gradnorm_tol=1e-4))
    X = opt.optimize(X0)
    res = opt.getOptResult()
    Xo, reso = Po.optimize(X0, RTR_iterations=outer, RTR_tCG_iterations=200, gradnorm_tol=1e-4)
    print(outer, "hip", res["fOpt"], res["gradNormOpt"], res["outer_iterations"], res["inner_iterations"], res["accepted_steps"], res["tCGStatus"],
          "| cpu", reso["fOpt"], reso["gradNormOpt"], reso["outer_iters"], reso["inner_iters"], reso["accepted"], reso["tcg_status"], flush=True)
for outer in (12, 19, 30, 60):
    opt = da.QuadraticOptimizer(P, da.ROptParameters(RTR_iterations=outer, RTR_tCG_iterations=200, gradnorm_tol=1e-4))
    t = time.time(); X = opt.optimize(X0); dt = time.time() - t
    res = opt.getOptResult()
    print(outer, "hip %.12f" % res["fOpt"], res["gradNormOpt"], res["outer_iterations"], res["inner_iterations"], res["accepted_steps"], res["tCGStatus"], "%.2fs" % dt, flush=True)
Xo, reso = Po.optimize(X0, RTR_iterations=200, RTR_tCG_iterations=200, gradnorm_tol=1e-4)
print("cpu %.12f" % reso["fOpt"], reso["gradNormOpt"], reso["outer_iters"], reso["inner_iters"], reso["accepted"], reso["tcg_status"])
