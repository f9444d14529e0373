"""the CORA staircase on tiers.pyfg with the time of every step of every level (problem, optimize, certificate, escape):
where config4_tiers.ms_to_certified_optimum goes.  python tools/c4_levels.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dcora_amd as da  # noqa: E402
from dcora_amd import cora_flow, datasets  # noqa: E402

ra = da.RADataset(os.path.join(datasets.DATA, "tiers.pyfg.gz"))


class Timed(cora_flow.ProductBackend):
    def __init__(self, ra):
        super().__init__(ra)
        self.t = {}

    def _t(self, name, f, *a):
        t0 = time.perf_counter()
        out = f(*a)
        self.t.setdefault(name, []).append(1e3 * (time.perf_counter() - t0))
        return out

    def problem(self, r):
        return self._t("problem", super().problem, r)

    def optimize(self, P, X):
        return self._t("optimize", super().optimize, P, X)

    def certificate(self, r, X):
        return self._t("certificate", super().certificate, r, X)

    def escape(self, Pn, X, theta, v):
        return self._t("escape", super().escape, Pn, X, theta, v)

    def project(self, X, r):
        return self._t("project", super().project, X, r)


for rep in range(2):
    hip = Timed(ra)
    out = cora_flow.cora(hip, ra.X_odom, ra.d)
    print("total %.0f ms, certified %s at r = %d" % (out["ms_total"], out["certified"], out["r_final"]))
    for k, v in hip.t.items():
        print("  %-12s %8.0f ms  %s" % (k, sum(v), " ".join("%.0f" % x for x in v)))
    print("  tCG", [lv["inner"] for lv in out["levels"]], "outer", [lv["outer"] for lv in out["levels"]], flush=True)
