"""a few applications of the central preconditioner of the 100k lattice (for rocprofv3 --kernel-trace)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402

ds = synth.lattice_se3()
Q = da.build_Q_pgo(ds)
P = da.QuadraticProblem(5, 3, ds.n, Q)
ms, nb = P.time_precond(reps=5)
print("apply %.1f us, %.1f MB" % (1e3 * ms, nb / 1e6), flush=True)
P.close()
