"""scratch benchmark (not a test): BASELINE.json config 5 -- synthetic 100k-pose SE(3) lattice, 8 agents, r = 5,
RBCD++ on one GPU; optional CPU oracle on a few iterations of the same run for the trace check and the ratio."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da  # noqa: E402
from dcora_amd import synth  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cpu_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    R, r = 8, 5
    t0 = time.time()
    ds = synth.lattice_se3()
    t_gen = time.time() - t0
    rng = np.random.default_rng(20250310)
    X0 = da.manifold_project(r, ds.d, ds.n, rng.uniform(-1, 1, (r, 4 * ds.n)))
    t0 = time.time()
    s = da.RbcdSession(ds, num_robots=R, r=r)
    t_setup = time.time() - t0
    s.set_X(X0)
    out = s.run(max_iters=3, rgrad_tol=0.0)  # warm-up
    s.set_X(X0)
    t0 = time.time()
    out = s.run(max_iters=iters, rgrad_tol=0.0)
    dt = time.time() - t0
    line = {"workload": "lattice 50x50x40 (100k poses, %d edges), %d agents, r=%d" % (ds.m, R, r),
            "gen_s": t_gen, "setup_s": t_setup, "iters": iters, "ms_per_iter": 1e3 * dt / iters,
            "it_per_s": iters / dt, "cost_first": float(out["cost"][0]), "cost_last": float(out["cost"][-1]),
            "gradnorm_last": float(out["gradnorm"][-1]), "selected": [int(x) for x in out["selected"][:12]]}
    print(json.dumps(line), flush=True)
    if cpu_iters:
        from oracle import orc
        dso = orc.Dataset(ds.d, ds.n, ds.ids, ds.vals)
        t0 = time.time()
        tr = orc.run_rbcd(dso, X0, num_robots=R, r_min=r, max_iters=cpu_iters, staircase=0, rgrad_tol=0.0)
        dtc = time.time() - t0
        n = min(cpu_iters, iters)
        line = {"cpu_iters": cpu_iters, "cpu_wall_s": dtc, "cpu_loop_s": tr["rbcd_seconds"],
                "cpu_ms_per_iter": 1e3 * tr["rbcd_seconds"] / tr["total_iters"],
                "selected_equal": bool(np.array_equal(tr["selected"][:n], out["selected"][:n])),
                "cost_rel_diff": float(np.max(np.abs(tr["cost"][:n] - out["cost"][:n]) / np.abs(tr["cost"][:n]))),
                "cpu_cost": [float(x) for x in tr["cost"][:n]], "gpu_cost": [float(x) for x in out["cost"][:n]]}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
