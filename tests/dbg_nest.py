import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dcora_amd as da
from dcora_amd import capi
from oracle import orc
L = capi.lib()
dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
L.dcora_debug_nesterov.argtypes = [C.c_int]*8 + [C.c_double]*2 + [dp]*6
r, d, n = 5, 3, 125
rng = np.random.default_rng(0)
k = 4*n
def run(fl, mode, restart, lo, hi, al, ga):
    rs = np.random.default_rng(1)
    A = [capi.F(orc.project_to_manifold(r,d,n, rs.standard_normal((r,k)))) for _ in range(6)]
    L.dcora_debug_nesterov(fl, r, d, n, mode, restart, lo, hi, al, ga, *A)
    return A
for mode in (0,1,2,3):
  for restart in (0,1):
    a = run(0, mode, restart, 25, 50, 0.37, 1.9)
    b = run(1, mode, restart, 25, 50, 0.37, 1.9)
    print(mode, restart, [float(np.abs(x-y).max()) for x,y in zip(a,b)])
rs = np.random.default_rng(1)
A0 = [capi.F(orc.project_to_manifold(r,d,n, rs.standard_normal((r,k)))) for _ in range(6)]
for fl in (0,1):
    a = run(fl, 0, 0, 25, 50, 0.37, 1.9)
    dv = np.abs(a[1]-A0[1]).reshape(n, 4, r)
    print('flavour', fl, 'V vs input max', dv.max(), 'poses with diff', np.where(dv.max(axis=(1,2))>1e-9)[0][:20], 'col', np.where(dv.max(axis=(0,2))>1e-9)[0])
a = run(1, 0, 0, 25, 50, 0.37, 1.9)
Vin = A0[1].reshape(n,4,r); Vout = a[1].reshape(n,4,r)
for p in (0, 11, 60):
    print('pose', p); print(np.round(Vin[p][:3].T,4)); print(np.round(Vout[p][:3].T,4)); print('VtV', np.round(Vout[p][:3]@Vout[p][:3].T,6))
dv = np.abs(a[1]-A0[1]).reshape(n, 4, r)
print(np.where(dv.max(axis=(1,2))>1e-9)[0])
print("=====")
for p in (0, 1):
    print('pose', p); print(np.round(Vin[p][:3].T,4)); print(np.round(Vout[p][:3].T,4)); print('Vin^T Vout', np.round(Vin[p][:3]@Vout[p][:3].T,6))
