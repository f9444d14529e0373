import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
n = 400
Z = da.Csr.from_scipy(sp.csr_matrix((np.zeros(n), (np.arange(n), np.arange(n))), shape=(n, n)))
print("zero:", da.min_eig(Z, tol=1e-6)[:2], da.min_eig(Z, tol=1e-6)[3])
I = da.Csr.from_scipy((2.0 * sp.identity(n)).tocsr())
print("2 I:", da.min_eig(I, tol=1e-6)[:2], da.min_eig(I, tol=1e-6)[3])
D = da.Csr.from_scipy(sp.diags(np.r_[-2.0, np.ones(200), 3.0 * np.ones(199)]).tocsr())
print("diag:", da.min_eig(D, tol=1e-8)[:2])
