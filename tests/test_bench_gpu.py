"""bench.py as the driver starts it: the line of the N = 1 run and of a 2-rank rehearsal (two processes on ONE GPU over
gloo -- not a scaling figure, a check that the SCALE path runs and prints a line the driver can parse)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert lines, p.stderr[-2000:]
    last = lines[-1]
    assert len(last) <= 8000
    return json.loads(last, parse_constant=lambda t: pytest.fail("non-finite constant " + t))


def test_headline_line_of_the_one_gpu_run():
    line = run_bench(["--steps", "20", "--warmup", "5", "--headline-only"])
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5 and line["value"] > 100
    assert abs(line["value"] * line["ms_per_step"] - 1e3) < 1.0
    assert line["config"]["workload"].startswith("sphere2500") and line["vs_baseline"] is None
    with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json")) as fh:
        detail = json.load(fh)
    assert abs(detail["value"] - line["value"]) <= 1e-5 * line["value"]


def test_two_rank_rehearsal_prints_the_scaling_series():
    line = run_bench(["--gpus", "2", "--steps", "10", "--warmup", "3", "--scaling-only"],
                     {"DCORA_DIST_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["value"] > 50
    sc = line["scaling_100k_lattice"]
    assert sc["agents"] == 16 and sc["n_gpus"] == 2 and sc["sweeps_per_s"] > 1, sc
    assert line["scaling_value"] == sc["sweeps_per_s"]
    assert sc["secondary_R_2N"]["agents"] == 4 and sc["secondary_R_2N"]["sweeps_per_s"] > 1
    assert line["process_group"]["backend"] == "gloo" and line["process_group"]["ranks"] == 2
