"""The line bench.py prints last on stdout must fit the driver's 8 KB tail window and parse as strict JSON (round 4's
line had grown to 22.5 KB and the driver recorded `parsed: null`).  compact_line() is a pure function of the dict the
stages build; it is run here on the committed full lines of earlier rounds and on degenerate inputs.  No GPU."""
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FULL_LINES = ["profiles/r04_bench.json", "profiles/r04_bench_2ranks_on_one_gpu_gloo.json", "profiles/r03_bench.json",
              "profiles/r02_bench.json"]
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def strict_loads(text):
    def refuse(token):
        raise ValueError("non-finite constant in the line: " + token)
    return json.loads(text, parse_constant=refuse)


@pytest.mark.parametrize("path", FULL_LINES)
def test_committed_full_lines_compact_to_a_parseable_line(path):
    with open(os.path.join(ROOT, path)) as fh:
        full = json.load(fh)
    line = bench.compact_line(full)
    text = json.dumps(line, allow_nan=False)
    assert len(text) <= bench.LINE_LIMIT and "\n" not in text
    back = strict_loads(text)
    assert back == line
    for key in CONTRACT_KEYS:
        assert key in back, key
    assert back["vs_baseline"] is None and back["dtype"] == "f64"
    assert "workload" in back["config"] and "model" not in back["config"]
    assert math.isclose(back["value"], full["value"], rel_tol=1e-6)
    assert math.isclose(back["value"] * back["ms_per_step"], 1e3, rel_tol=1e-4)   # the driver's consistency check
    rf, cb = back["roofline"], back["cpu_baseline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and "traffic" in rf
    assert math.isclose(rf["frac"], rf["achieved"] / rf["peak"], rel_tol=1e-5)
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    for text_field in (back["config"]["window_note"], rf["note"], cb["sample"]):
        assert len(text_field) <= 200


def test_the_line_sheds_optional_blocks_rather_than_grow():
    with open(os.path.join(ROOT, FULL_LINES[0])) as fh:
        full = json.load(fh)
    # a pathological run: every optional block carries a huge error string, every note is long
    for key in list(full):
        if isinstance(full[key], dict) and key not in ("config", "roofline", "cpu_baseline"):
            full[key]["error"] = "x" * 5000
    full["roofline_qapply"] = {"entry%d" % i: {"frac": 0.1 * i, "avg_launch_us": 1.0 + i, "kernel": "k" * 400}
                               for i in range(200)}
    line = bench.compact_line(full)
    text = json.dumps(line, allow_nan=False)
    assert len(text) <= bench.LINE_LIMIT
    for key in CONTRACT_KEYS + ("roofline", "cpu_baseline"):
        assert key in line


def test_headline_only_and_non_finite_inputs():
    full = {"metric": "m", "value": 10.0, "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 100.0,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "d",
            "config": {"workload": "w"}}
    line = bench.compact_line(full)
    assert strict_loads(json.dumps(line, allow_nan=False))["vs_baseline"] is None
    # emit() cleans NaN / Infinity / numpy scalars before the line is built
    import io
    import numpy as np
    full["config"]["final_cost_2f"] = float("nan")
    full["sustained"] = {"value": np.float64(3.5), "steps": np.int64(7), "ms_per_step": float("inf")}
    out = io.StringIO()
    saved = bench.DETAIL_PATH
    bench.DETAIL_PATH = os.path.join(ROOT, "gpurun_out", "test_bench_detail.json")
    try:
        bench.emit(out, full)
    finally:
        bench.DETAIL_PATH = saved
    lines = out.getvalue().splitlines()
    assert len(lines) == 1
    back = strict_loads(lines[0])
    assert back["sustained"]["value"] == 3.5 and "ms_per_step" not in back["sustained"]
    assert "final_cost_2f" not in back["config"]


def test_scaling_block_names_one_partition_at_every_n():
    one = {"one_gpu": {"R=16": {"sweeps_per_s": 150.0, "block_updates_per_s": 2400.0}, "R=4": {"sweeps_per_s": 151.0},
                       "R=8": {"sweeps_per_s": 190.0}}}
    b1 = bench.scaling_block(1, one, {"value": 1450.0, "unit": "RBCD iterations/s"}, None)
    assert b1["compact"]["agents"] == 16 and b1["compact"]["sweeps_per_s"] == 150.0
    cached = {"strong_one_gpu": one["one_gpu"], "provenance": {"git_head": "abc", "age_s": 10.0}}
    multi = {"R=16": {"sweeps_per_s": 450.0, "block_updates_per_s": 7200.0}, "R=8": {"sweeps_per_s": 400.0}}
    b4 = bench.scaling_block(4, multi, None, cached)
    assert b4["compact"]["agents"] == 16 and b4["compact"]["speedup_vs_one_gpu"] == 3.0
    assert b4["compact"]["secondary_R_2N"] == {"agents": 8, "sweeps_per_s": 400.0, "one_gpu_sweeps_per_s": 190.0}
    b4n = bench.scaling_block(4, multi, None, None, "N = 1 cache ignored: git_head differs")
    assert b4n["compact"]["one_gpu_sweeps_per_s"] is None and "ignored" in b4n["compact"]["one_gpu_note"]
    full = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 4, "steps": 1, "warmup": 0, "ms_per_step": 1000.0,
            "config": {"workload": "w"}, "scaling_100k_lattice": b4}
    line = bench.compact_line(full)
    assert line["scaling_value"] == 450.0 and line["scaling_100k_lattice"]["n_gpus"] == 4


def test_n1_cache_is_ignored_on_any_provenance_mismatch(tmp_path, monkeypatch):
    class Args:
        dataset, rank_r, robots, steps, warmup = "sphere2500", 5, 5, 20, 5
    monkeypatch.setattr(bench, "CACHE_PATH", str(tmp_path / "cache.json"))
    assert bench.cache_load(Args)[0] is None
    bench.cache_store({"cpu_baseline": {"value": 80.0}, "provenance": bench.provenance(Args)})
    c, why = bench.cache_load(Args)
    assert why is None and c["cpu_baseline"]["value"] == 80.0 and c["provenance"]["age_s"] >= 0

    class Other(Args):
        robots = 4
    c, why = bench.cache_load(Other)
    assert c is None and "robots" in why
    stale = dict(c=1)
    stale = {"cpu_baseline": {"value": 80.0}, "provenance": dict(bench.provenance(Args), lib_sha16="0" * 16)}
    bench.cache_store(stale)
    c, why = bench.cache_load(Args)
    assert c is None and "lib_sha16" in why
