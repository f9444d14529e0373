"""creation of the dense-preconditioner problems of sphere2500 / 5 agents (cold cache), device against host factor"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dcora_amd as da  # noqa: E402
from dcora_amd import datasets  # noqa: E402

ds = datasets.product_dataset("sphere2500")
for R in (5, 2):
    nb, ids, vals = bench.agent_block(ds, R, 0)
    Q = da.build_Q_pgo(ds, n=nb, agent=0, ids=ids, vals=vals)
    for rep in range(3):
        da.precond_cache_clear()
        t = time.perf_counter()
        P = da.QuadraticProblem(5, 3, nb, Q)
        dt = time.perf_counter() - t
        rng = np.random.default_rng(1)
        X = da.manifold_project(5, 3, nb, rng.uniform(-1, 1, (5, 4 * nb)))
        V = rng.standard_normal((5, 4 * nb))
        Z = P.PreCondition(X, V)
        print("k %d create %.2f ms, |PreCondition| %.12e" % (4 * nb, 1e3 * dt, np.linalg.norm(Z)), flush=True)
        P.close()
for rep in range(2):
    da.precond_cache_clear()
    t = time.perf_counter()
    s = da.RbcdSession(ds, num_robots=5, r=5)
    print("session of 5: %.2f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
    s.close()
