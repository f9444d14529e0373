import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da
from dcora_amd import cora_flow, datasets
ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))
hip = cora_flow.ProductBackend(ra)
P = hip.problem(ra.d)
print(P.precond_info())
ms, by = P.time_precond(reps=200)
print("application %.1f us, %.1f MB" % (ms * 1e3, by / 1e6))
