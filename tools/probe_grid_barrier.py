"""cost of a grid-wide barrier among co-resident workgroups (dcora_debug_grid_barrier): what a persistent one-launch
tCG iteration would pay per dependency"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dcora_amd import capi  # noqa: E402

L = capi.lib()
L.dcora_debug_grid_barrier.restype = C.c_int
L.dcora_debug_grid_barrier.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double),
                                       C.POINTER(C.c_int)]
for mode in (0, 1):
    for blocks in (64, 128, 250, 256):
        us, to = C.c_double(), C.c_int()
        rc = L.dcora_debug_grid_barrier(0, blocks, 2000, 128 * 1024, mode, C.byref(us), C.byref(to))
        print("%s, %3d workgroups: rc %d, %.2f us per barrier, timeouts %d" % (
            "two levels" if mode else "one counter", blocks, rc, us.value, to.value), flush=True)
