"""scratch: BASELINE config 4 end to end -- the centralised CORA staircase on tiers.pyfg (odometry start -> certified ->
rounded), the flow of ref examples/SingleRobotExample_RASLAM.cpp:188-283, on the product"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dcora_amd as da
from dcora_amd import cora_flow, datasets
name = sys.argv[1] if len(sys.argv) > 1 else "tiers"
ra = da.RADataset(os.path.join(datasets.DATA, name + ".pyfg.gz"))
hip = cora_flow.ProductBackend(ra)
t0 = time.perf_counter()
out = cora_flow.cora(hip, ra.X_odom, ra.d, log=lambda lv: print(json.dumps(lv), flush=True))
print(json.dumps({"certified": out["certified"], "r_final": out["r_final"], "f_rounded": out["f_rounded"],
                  "ms_total": out["ms_total"], "levels": out["levels"]}))
