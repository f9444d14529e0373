"""creation of the whole 100k lattice as one problem (DCORA_INIT_TIMING laps on stderr; DCORA_HOST_THREADS caps the
host threads of the set-up)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402

ds = synth.lattice_se3()
Q = da.build_Q_pgo(ds)
t = time.perf_counter()
P = da.QuadraticProblem(5, 3, ds.n, Q)
print("threads %s: central create %.2f s" % (os.environ.get("DCORA_HOST_THREADS", "default"), time.perf_counter() - t),
      flush=True)
P.close()
